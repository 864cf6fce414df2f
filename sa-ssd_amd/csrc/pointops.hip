// pointops.hip -- training-side point operators of the auxiliary network:
//   three_nn / three_interpolate (+grad)   mmdet/ops/pointnet2/src/interpolate_gpu.cu:9-146 (wrappers
//                                          interpolate.cpp:13,25,40, used by necks/cmn.py:175-189)
//   pts_in_boxes3d                         mmdet/ops/points_op/src/points_op.cpp:92-144 (a serial CPU loop after a
//                                          .cpu() sync in the reference, cmn.py:48-54)
#include "common.h"

namespace {

constexpr int kNnTile = 1024;     // known points staged per LDS tile (16 KB)

// One thread per unknown point; known points stream through LDS in ascending order, so the strict '<' updates give
// exactly the reference's first-index-wins top-3 (interpolate_gpu.cu:29-50).
__global__ void __launch_bounds__(256) three_nn_kernel(int n, int m, const float *__restrict__ unknown,
                                                       const float *__restrict__ known, float *__restrict__ dist2,
                                                       int *__restrict__ idx)
{
#pragma clang fp contract(off)      // same IEEE fp32 operation sequence as the CPU oracle (squared distances bit-exact)
    __shared__ float4 tile[kNnTile];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < n) u = ((const float4 *)unknown)[p];
    double b1 = 1e40, b2 = 1e40, b3 = 1e40;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int t0 = 0; t0 < m; t0 += kNnTile) {
        const int cnt = min(kNnTile, m - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) tile[i] = ((const float4 *)known)[t0 + i];
        __syncthreads();
        if (p < n) {
            for (int k = 0; k < cnt; ++k) {
                const float4 q = tile[k];
                if (q.x != u.x) continue;
                const float d = (u.y - q.y) * (u.y - q.y) + (u.z - q.z) * (u.z - q.z) + (u.w - q.w) * (u.w - q.w);
                if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = t0 + k; }
                else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = t0 + k; }
                else if (d < b3) { b3 = d; i3 = t0 + k; }
            }
        }
    }
    if (p < n) {
        dist2[p * 3 + 0] = (float)b1; dist2[p * 3 + 1] = (float)b2; dist2[p * 3 + 2] = (float)b3;
        idx[p * 3 + 0] = i1; idx[p * 3 + 1] = i2; idx[p * 3 + 2] = i3;
    }
}

// ---- exact 3-NN through a coarse BEV binning of the known points ---------------------------------------------------
// The reference brute-forces O(N*M) (interpolate_gpu.cu:9-56).  Here the known points are counting-sorted into a
// uniform (batch, y, x) grid of `cell`-sized columns (all z), and a query walks Chebyshev rings of cells around its
// own cell, scoring every point in them.  Exactness: a point in a cell outside ring r differs from the query by
// more than r*cell in x or y (cell indices are clamped into the grid, which only makes the test more conservative),
// so once the 3rd best squared distance is <= (r*cell)^2 nothing farther can displace it; isolated points simply
// keep widening until the whole grid is covered.  Candidates are ranked by (distance, row) -- exactly the order the
// reference's ascending scan with strict '<' produces -- and distances use the same fp32 operation sequence, so
// dist2 / idx are bit-identical to sassd_three_nn.
struct NnBins {
    int nx, ny, nb;
    float x0, y0, cell;
};

__device__ __forceinline__ int nn_cell(const NnBins &g, float b, float x, float y, int *cx, int *cy)
{
    const int ib = (int)b;
    if (!(ib >= 0 && ib < g.nb && (float)ib == b)) return -1;
    *cx = min(max((int)floorf((x - g.x0) / g.cell), 0), g.nx - 1);
    *cy = min(max((int)floorf((y - g.y0) / g.cell), 0), g.ny - 1);
    return (ib * g.ny + *cy) * g.nx + *cx;
}

__global__ void nn_count_kernel(int m, const float *__restrict__ known, NnBins g, int *__restrict__ count)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const float4 q = ((const float4 *)known)[k];
    int cx, cy;
    const int c = nn_cell(g, q.x, q.y, q.z, &cx, &cy);
    if (c >= 0) atomicAdd(&count[c], 1);
}

// start[c] = exclusive prefix of count, cursor[c] = start[c], in two coalesced passes over 1024-cell blocks (a single
// block walking ~70 cells per thread was latency-bound at 60-150 us): pass 1 scans inside each block and records the
// block totals, pass 2 adds the totals of the preceding blocks (at most 1024 of them: one per thread).
__global__ void __launch_bounds__(1024) nn_scan_local_kernel(int ncell, const int *__restrict__ count,
                                                             int *__restrict__ start, int *__restrict__ btot)
{
    __shared__ int wsum[17];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int v = i < ncell ? count[i] : 0;
    int total;
    const int ex = block_exclusive_scan(v, wsum, &total);
    if (i < ncell) start[i] = ex;
    if (threadIdx.x == 0) btot[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024) nn_scan_offset_kernel(int ncell, const int *__restrict__ btot,
                                                              int *__restrict__ start, int *__restrict__ cursor)
{
    __shared__ int wsum[17];
    const int mine = (int)threadIdx.x < (int)blockIdx.x ? btot[threadIdx.x] : 0;      // blocks before this one
    const int last = (int)threadIdx.x < (int)gridDim.x ? btot[threadIdx.x] : 0;       // all blocks (for the grand total)
    int before, all;
    block_exclusive_scan(mine, wsum, &before);
    __syncthreads();
    block_exclusive_scan(last, wsum, &all);
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < ncell) {
        const int s = start[i] + before;
        start[i] = s;
        cursor[i] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) start[ncell] = all;
}

__global__ void nn_fill_kernel(int m, const float *__restrict__ known, NnBins g, int *__restrict__ cursor,
                               float4 *__restrict__ sorted, int *__restrict__ sorted_row)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const float4 q = ((const float4 *)known)[k];
    int cx, cy;
    const int c = nn_cell(g, q.x, q.y, q.z, &cx, &cy);
    if (c < 0) return;
    const int pos = atomicAdd(&cursor[c], 1);
    sorted[pos] = q;
    sorted_row[pos] = k;
}

// three best (distance, row) pairs under the total order (d, row): the result does not depend on the visiting order
struct Top3 {
    float b1, b2, b3;                                       // +inf = empty (the reference prints 1e40 -> inf as float)
    int i1, i2, i3;
    __device__ __forceinline__ void init() { b1 = b2 = b3 = __builtin_huge_valf(); i1 = i2 = i3 = 0; }
    __device__ __forceinline__ static bool less(float d, int i, float bd, int bi)
    {
        return d < bd || (d == bd && i < bi);
    }
    __device__ __forceinline__ void push(float d, int i)
    {
        if (less(d, i, b1, i1)) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = i; }
        else if (less(d, i, b2, i2)) { b3 = b2; i3 = i2; b2 = d; i2 = i; }
        else if (less(d, i, b3, i3)) { b3 = d; i3 = i; }
    }
    __device__ __forceinline__ bool has(float d, int i) const
    {
        return (d == b1 && i == i1) || (d == b2 && i == i2) || (d == b3 && i == i3);
    }
    // union with the partner lane's list (both lanes end with the same three).  After the first ring the lists of a group
    // share the entries of the previous merge: a candidate (= a row index) already present is not inserted twice.
    __device__ __forceinline__ void merge_xor(int mask)
    {
        const float p1 = __shfl_xor(b1, mask, 64), p2 = __shfl_xor(b2, mask, 64), p3 = __shfl_xor(b3, mask, 64);
        const int j1 = __shfl_xor(i1, mask, 64), j2 = __shfl_xor(i2, mask, 64), j3 = __shfl_xor(i3, mask, 64);
        if (p1 < __builtin_huge_valf() && !has(p1, j1)) push(p1, j1);
        if (p2 < __builtin_huge_valf() && !has(p2, j2)) push(p2, j2);
        if (p3 < __builtin_huge_valf() && !has(p3, j3)) push(p3, j3);
    }
};

// FOUR lanes per unknown point (a thread per point walked ~130 candidates serially with 64 different trip counts per
// wave, and 32 k points filled half the chip: 212 us per call): lane `sub` of a group takes every fourth candidate of a
// cell, the four partial lists are merged by two xor-shuffles after every ring, so the stop test and the result are those
// of the serial search (the (distance, row) order is total: any visiting order gives the same three).
__global__ void __launch_bounds__(256) three_nn_binned_kernel(int n, const float *__restrict__ unknown, NnBins g,
                                                              const int *__restrict__ start,
                                                              const float4 *__restrict__ sorted,
                                                              const int *__restrict__ sorted_row,
                                                              float *__restrict__ dist2, int *__restrict__ idx)
{
#pragma clang fp contract(off)
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int p = gt >> 2, sub = gt & 3;
    if (p >= n) return;                                     // (whole groups leave together)
    const float4 u = ((const float4 *)unknown)[p];
    Top3 t;
    t.init();
    int cx, cy;
    const int c0 = nn_cell(g, u.x, u.y, u.z, &cx, &cy);
    if (c0 >= 0) {
        const int base = (int)u.x * g.ny;
        const int rmax = max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy));
        for (int r = 0; r <= rmax; ++r) {
            for (int y = cy - r; y <= cy + r; ++y) {
                if (y < 0 || y >= g.ny) continue;
                const bool edge_row = (y == cy - r) || (y == cy + r);
                const int step = edge_row ? 1 : max(2 * r, 1);           // interior rows: only the two end cells
                for (int x = cx - r; x <= cx + r; x += step) {
                    if (x < 0 || x >= g.nx) continue;
                    const int c = (base + y) * g.nx + x;
                    const int e = start[c + 1];
                    for (int j = start[c] + sub; j < e; j += 4) {
                        const float4 q = sorted[j];
                        const float d = (u.y - q.y) * (u.y - q.y) + (u.z - q.z) * (u.z - q.z) +
                                        (u.w - q.w) * (u.w - q.w);
                        t.push(d, sorted_row[j]);
                    }
                }
            }
            t.merge_xor(1);
            t.merge_xor(2);
            const float reach = (float)r * g.cell * 0.999f;             // 0.1 % slack for cell-boundary rounding
            if (r >= 1 && (double)t.b3 <= (double)reach * (double)reach) break;
        }
    }
    if (sub == 0) {
        dist2[p * 3 + 0] = t.b1; dist2[p * 3 + 1] = t.b2; dist2[p * 3 + 2] = t.b3;
        idx[p * 3 + 0] = t.i1; idx[p * 3 + 1] = t.i2; idx[p * 3 + 2] = t.i3;
    }
}

// one thread per (point, channel); consecutive threads -> consecutive channels (coalesced rows)
__global__ void three_interpolate_kernel(int c, int m, int n, const float *__restrict__ points,
                                         const int *__restrict__ idx, const float *__restrict__ weight,
                                         float *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * c) return;
    const int p = t / c, ch = t - (size_t)p * c;
    const int *id = idx + p * 3;
    const float *w = weight + p * 3;
    out[t] = w[0] * points[(size_t)id[0] * c + ch] + w[1] * points[(size_t)id[1] * c + ch] +
             w[2] * points[(size_t)id[2] * c + ch];
}

__global__ void three_interpolate_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                              const int *__restrict__ idx, const float *__restrict__ weight,
                                              float *__restrict__ grad_points)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * c) return;
    const int p = t / c, ch = t - (size_t)p * c;
    const int *id = idx + p * 3;
    const float *w = weight + p * 3;
    const float g = grad_out[t];
    atomicAdd(grad_points + (size_t)id[0] * c + ch, g * w[0]);
    atomicAdd(grad_points + (size_t)id[1] * c + ch, g * w[1]);
    atomicAdd(grad_points + (size_t)id[2] * c + ch, g * w[2]);
}

// points_op.cpp:92-105 -- note the literal argument order of the call site (:130-133): w = box[3], l = box[4],
// h = box[5]; double-precision halves exactly as the C++ (x / 2.0 promotes to double).
__device__ __forceinline__ int pt_in_box3d(float x, float y, float z, float cx, float cy, float bottom_z, float w,
                                           float l, float h, float angle)
{
    const float max_dis = 10.0f;
    const float cz = (float)((double)bottom_z + (double)h / 2.0);
    if ((fabsf(x - cx) > max_dis) || ((double)fabsf(z - cz) > (double)h / 2.0) || (fabsf(y - cy) > max_dis)) return 0;
    const float cosa = cosf(angle), sina = sinf(angle);
    const float x_rot = (x - cx) * cosa + (y - cy) * (-sina);
    const float y_rot = (x - cx) * sina + (y - cy) * cosa;
    return ((double)x_rot >= -(double)w / 2.0) & ((double)x_rot <= (double)w / 2.0) &
           ((double)y_rot >= -(double)l / 2.0) & ((double)y_rot <= (double)l / 2.0);
}

// one thread per point, boxes in ascending order: the LAST containing box defines the centre offset (:135-139)
__global__ void pts_in_boxes3d_kernel(const float *__restrict__ pts, int n, const float *__restrict__ boxes, int m,
                                      int32_t *__restrict__ flag, float *__restrict__ reg)
{
#pragma clang fp contract(off)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float x = pts[j * 3], y = pts[j * 3 + 1], z = pts[j * 3 + 2];
    for (int i = 0; i < m; ++i) {
        const float *b = boxes + i * 7;
        const int in = pt_in_box3d(x, y, z, b[0], b[1], b[2], b[3], b[4], b[5], b[6]);
        flag[(size_t)i * n + j] = in;
        if (in == 1) {
            reg[j * 3] = x - b[0];
            reg[j * 3 + 1] = y - b[1];
            reg[j * 3 + 2] = (float)((double)z - ((double)b[2] + (double)b[3] / 2.0));
        }
    }
}

}  // namespace

extern "C" int sassd_three_nn(int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx,
                              void *stream_)
{
    if (n < 0 || m < 0 || !dist2 || !idx) return SASSD_EINVAL;
    if (n == 0) return SASSD_OK;
    if (!unknown || (m > 0 && !known)) return SASSD_EINVAL;
    hipLaunchKernelGGL(three_nn_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream_, n, m, unknown, known,
                       dist2, idx);
    return sassd_launch_status();
}

namespace {
struct NnLayout { size_t count, start, cursor, btot, sorted, rows, total; int ncell; };
NnLayout nn_layout(int m, int nx, int ny, int nb)
{
    NnLayout L;
    L.ncell = nx * ny * nb;
    size_t o = 0;
    L.count = o;  o = align_up(o + (size_t)L.ncell * 4, 256);
    L.start = o;  o = align_up(o + (size_t)(L.ncell + 1) * 4, 256);
    L.cursor = o; o = align_up(o + (size_t)L.ncell * 4, 256);
    L.btot = o;   o = align_up(o + (size_t)1024 * 4, 256);
    L.sorted = o; o = align_up(o + (size_t)(m > 0 ? m : 1) * 16, 256);
    L.rows = o;   o = align_up(o + (size_t)(m > 0 ? m : 1) * 4, 256);
    L.total = o;
    return L;
}
bool nn_grid_ok(int nx, int ny, int nb) { return nx >= 1 && ny >= 1 && nb >= 1 && (long)nx * ny * nb <= (1 << 20); }
}  // namespace

extern "C" size_t sassd_three_nn_binned_workspace_bytes(int m, int nx, int ny, int batch_size)
{
    return nn_grid_ok(nx, ny, batch_size) && m >= 0 ? nn_layout(m, nx, ny, batch_size).total : 0;
}

extern "C" int sassd_three_nn_binned(int n, int m, const float *unknown, const float *known, float x0, float y0,
                                     float cell, int nx, int ny, int batch_size, float *dist2, int32_t *idx,
                                     void *workspace, size_t workspace_bytes, void *stream_)
{
    if (n < 0 || m < 0 || !dist2 || !idx || !workspace || !(cell > 0.f) || !nn_grid_ok(nx, ny, batch_size))
        return SASSD_EINVAL;
    const NnLayout L = nn_layout(m, nx, ny, batch_size);
    if (workspace_bytes < L.total) return SASSD_ENOSPC;
    if (n == 0) return SASSD_OK;
    if (!unknown || (m > 0 && !known)) return SASSD_EINVAL;
    hipStream_t s = (hipStream_t)stream_;
    char *ws = (char *)workspace;
    int *count = (int *)(ws + L.count), *start = (int *)(ws + L.start), *cursor = (int *)(ws + L.cursor);
    float4 *sorted = (float4 *)(ws + L.sorted);
    int *rows = (int *)(ws + L.rows);
    NnBins g;
    g.nx = nx; g.ny = ny; g.nb = batch_size; g.x0 = x0; g.y0 = y0; g.cell = cell;
    if (hipMemsetAsync(count, 0, (size_t)L.ncell * 4, s) != hipSuccess) return sassd_launch_status();
    if (m > 0) hipLaunchKernelGGL(nn_count_kernel, dim3(cdiv(m, 256)), dim3(256), 0, s, m, known, g, count);
    int *btot = (int *)(ws + L.btot);
    const int nsb = cdiv(L.ncell, 1024);                       // <= 1024 (nn_grid_ok)
    hipLaunchKernelGGL(nn_scan_local_kernel, dim3(nsb), dim3(1024), 0, s, L.ncell, count, start, btot);
    hipLaunchKernelGGL(nn_scan_offset_kernel, dim3(nsb), dim3(1024), 0, s, L.ncell, (const int *)btot, start, cursor);
    if (m > 0) hipLaunchKernelGGL(nn_fill_kernel, dim3(cdiv(m, 256)), dim3(256), 0, s, m, known, g, cursor, sorted, rows);
    hipLaunchKernelGGL(three_nn_binned_kernel, dim3(cdiv(n, 64)), dim3(256), 0, s, n, unknown, g, start, sorted, rows,
                       dist2, idx);
    return sassd_launch_status();
}

extern "C" int sassd_three_interpolate(int c, int m, int n, const float *points, const int32_t *idx,
                                       const float *weight, float *out, void *stream_)
{
    if (c < 1 || m < 0 || n < 0 || !idx || !weight || !out) return SASSD_EINVAL;
    if (n == 0) return SASSD_OK;
    if (!points) return SASSD_EINVAL;
    const size_t tot = (size_t)n * c;
    hipLaunchKernelGGL(three_interpolate_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream_, c, m, n, points, idx, weight, out);
    return sassd_launch_status();
}

extern "C" int sassd_three_interpolate_grad(int c, int n, int m, const float *grad_out, const int32_t *idx,
                                            const float *weight, float *grad_points, void *stream_)
{
    if (c < 1 || m < 0 || n < 0 || !idx || !weight || !grad_points) return SASSD_EINVAL;
    if (n == 0) return SASSD_OK;
    if (!grad_out) return SASSD_EINVAL;
    const size_t tot = (size_t)n * c;
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream_, c, n, m, grad_out, idx, weight, grad_points);
    return sassd_launch_status();
}

extern "C" int sassd_pts_in_boxes3d(const float *pts, int n, const float *boxes3d, int m, int32_t *pts_flag,
                                    float *reg_target, void *stream_)
{
    if (n < 0 || m < 0 || !pts_flag || !reg_target) return SASSD_EINVAL;
    if (n == 0 || m == 0) return SASSD_OK;
    if (!pts || !boxes3d) return SASSD_EINVAL;
    hipLaunchKernelGGL(pts_in_boxes3d_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream_, pts, n, boxes3d,
                       m, pts_flag, reg_target);
    return sassd_launch_status();
}
