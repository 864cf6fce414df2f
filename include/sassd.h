/*
 * sassd.h -- C ABI of the MI355X-native SA-SSD hot path (libsassd.so, gfx950 only).
 *
 * This is the drop-in boundary.  Every entry point replaces one operator surface of the reference
 * (skyhehe123/SA-SSD; file:line relative to /root/reference) -- the reference binds those through pybind11 /
 * numba / the spconv Python package; a maintainer binds these through ctypes.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, a `void* stream` that is a hipStream_t (NULL = default).
 *   - return 0 on success, negative SASSD_E* otherwise.  No function allocates, frees or synchronises:
 *     the caller owns every buffer (use the *_workspace_bytes queries) and every count that is data
 *     dependent lives in DEVICE memory (int32 scalars), so a whole frame is a fixed launch sequence
 *     (hipGraph-capturable).  The reference, by contrast, exit()s on CUDA errors (iou3d.cpp:13-21) and
 *     syncs on every boolean index (ssd_rotate_head.py:336-356).
 *   - "cap" arguments are row capacities of caller buffers; if a device count exceeds its capacity the
 *     kernel clamps, and sets bit i of the int32 at `status` (see SASSD_ST_*) when a status pointer is given.
 *   - all floating point is IEEE fp32; all index math int32 (linear voxel indices must fit 2^32-2).
 */
#ifndef SASSD_H
#define SASSD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SASSD_OK       0
#define SASSD_EINVAL  -1   /* bad argument / unsupported configuration   */
#define SASSD_ENOSPC  -2   /* caller workspace or capacity too small     */
#define SASSD_EHIP    -3   /* HIP runtime / launch error (sassd_last_hip_error) */

#define SASSD_ST_VOXEL_OVERFLOW   1   /* more rows than an output capacity      */
#define SASSD_ST_HASH_FULL        2   /* hash table probe exhausted             */
#define SASSD_ST_BOX_OVERFLOW     4   /* more candidate boxes than capK / capD  */
#define SASSD_ST_GRID_SYNC        8   /* an in-launch grid barrier timed out (persistent rulebook pyramid) */

#define SASSD_MAX_POINTS_PER_VOXEL 64  /* max_points supported by sassd_voxelize (reference default: 35) */

const char *sassd_version(void);
int sassd_last_hip_error(void);
const char *sassd_last_hip_error_string(void);

/* Kernel selection is PER CALL (round 6; rounds 1-5 had process-wide sassd_debug_set_* switches, which a hipGraph capture
 * could bake in by accident -- none is left): the sparse-conv and Winograd entry points take an `int cfg`, 0 in production; the
 * direct and the bf16 convolution have `_cfg` twins of their entry points for their ablation switches.
 *   sparse conv (sassd_spconv_fwd / _bwd_data / _bwd_weight)   low 16 bits = ablation flags (tools/ablate_spconv.py): bit2 no
 *     MFMA, bit5 every gather reads row 0 (weight gradient: the tile-per-wave formulation), bit6 one weight image for every
 *     offset (bits 5 / 6 keep the number of loads in flight unchanged), bit8 the register-stationary kernel everywhere.
 *     Bits 16+: workgroup geometry -- 0 the per-shape, per-capacity default; 1 / 5 = spconv_gs_kernel on 4 / 8 waves, 8 / 9 =
 *     the balanced kernel spconv_gq_kernel with 4x4x1 quads / 16x16x4 tiles (the four the default picks from), 10 = the
 *     round-3 default (1 or 5 by capacity) for every layer; any other value returns SASSD_EINVAL.
 *   Winograd (sassd_conv2d_wino4_fwd / _chain, sassd_conv1x1_gemm_fwd)   bits 0-7 = GEMM geometry: 0 split operands on the
 *     bf16 MFMA, 128 x 128 workgroups (default); 1 = the fp32 MFMA (v_mfma_f32_32x32x2_f32) at its picked width; 2..6 = fp32
 *     MFMA with 32 cfg tile columns; 11..14 = split with other workgroup shapes.  Bits 8+: bit0 stage only the first chunk,
 *     bit1 no MFMA (fp32 geometries), bit2 no split arithmetic, bits 4 / 5 / 6 / 7 skip the input transform / GEMM / output
 *     transform / fused transform (per-kernel timing on live buffers, bench.py).  Chained calls must agree on the geometry.
 * The Python binding reads SASSD_SPCONV_DEBUG / SASSD_WINO4_CFG once as the DEFAULT cfg its wrappers pass. */

/* hipGraph capture of a launch sequence issued through this ABI (the reference has no counterpart: its frame is
 * ~10^2 host-issued launches with >= 6 host syncs).  begin -> any sassd_* calls on `stream` (streams
 * joined to it through events are captured too) -> end returns an executable graph; launch replays it. */
int sassd_graph_begin(void *stream);
int sassd_graph_end(void *stream, void **graph_exec);
int sassd_graph_launch(void *graph_exec, void *stream);
int sassd_graph_destroy(void *graph_exec);

/* ------------------------------------------------------------------------------------------------
 * (a1+a2) Hard voxelisation + per-voxel mean.
 * Replaces mmdet/ops/points_op/points_ops.py:104-164 `points_to_voxel` (numba kernel :5-50, zyx order) and
 * mmdet/models/backbones/vxnet.py:110-116 `SimpleVoxel.forward`.  Bit-exact with the serial reference:
 * voxels in first-touch order, <= max_points points per voxel in arrival order, the max_voxels `break`.
 *   points      [n_points, ndim] f32 (ndim >= 3; xyz first)            device
 *   voxel_size  3 f32, coors_range 6 f32                               HOST
 *   voxels      [cap, max_points, ndim] f32 zero padded, or NULL
 *   coors       [cap, coors_cols] i32; coors_cols 3 -> (z,y,x); 4 -> (batch_idx,z,y,x)
 *   num_points  [cap] i32, or NULL
 *   mean        [cap, nfeat] f32 (nfeat <= ndim), or NULL
 *   row_offset  device int32[2] or NULL: rows are written starting at row_offset[0] (0 if NULL) and
 *               row_offset[1] = row_offset[0] + voxel_num is written back (batch concatenation,
 *               detectors/single_stage.py:52-73 merge_second_batch).
 *   voxel_num   device int32 scalar: number of voxels of THIS cloud
 *   workspace   16-byte aligned (its tables are cleared by 16-byte stores; SASSD_EINVAL otherwise)
 * ---------------------------------------------------------------------------------------------- */
size_t sassd_voxelize_workspace_bytes(int n_points, int max_points);
int sassd_voxelize(const float *points, int n_points, int ndim, const float *voxel_size,
                   const float *coors_range, int max_points, int max_voxels, int batch_idx,
                   float *voxels, int32_t *coors, int coors_cols, int32_t *num_points, float *mean,
                   int nfeat, int32_t *row_offset, int32_t *voxel_num, int cap, int32_t *status,
                   void *workspace, size_t workspace_bytes, void *stream);

/* The same with the point count in DEVICE memory: `points` has room for points_cap rows, min(*n_points_dev, points_cap)
 * of them are valid.  The launch sequence depends on points_cap only, so a frame that starts at the raw point cloud
 * can be captured in a hipGraph and replayed for clouds of any size (workspace: sassd_voxelize_workspace_bytes(points_cap, ..)). */
int sassd_voxelize_dev(const float *points, int points_cap, const int32_t *n_points_dev, int ndim,
                       const float *voxel_size, const float *coors_range, int max_points, int max_voxels,
                       int batch_idx, float *voxels, int32_t *coors, int coors_cols, int32_t *num_points,
                       float *mean, int nfeat, int32_t *row_offset, int32_t *voxel_num, int cap, int32_t *status,
                       void *workspace, size_t workspace_bytes, void *stream);

/* SimpleVoxel.forward alone (vxnet.py:110-116): mean[v,f] = sum_t voxels[v,t,f] / num_points[v]. */
int sassd_voxel_mean(const float *voxels, const int32_t *num_points, int m, int max_points, int ndim,
                     int nfeat, float *mean, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (a5) Rulebook build.  Replaces spconv v1.0 `get_indice_pairs` as invoked by SubMConv3d / SparseConv3d
 * (call sites mmdet/models/necks/cmn.py:147-173).  The rulebook is a gather table
 *   nbr[row_out, 27]  = input row feeding output row_out through kernel offset k=(kz*3+ky)*3+kx, or -1
 * (each (out,k) has at most one input); sassd_rulebook_pairs converts it to spconv's
 * indice_pairs[27,2,P] / indice_pair_num[27] form (pairs in ascending out row).
 *   indices [n,4] i32 (batch,z,y,x); n_ptr device int32 row count; cap = row capacity of indices/nbr.
 * ---------------------------------------------------------------------------------------------- */
size_t sassd_hash_bytes(int cap_rows);                 /* table for cap_rows keys                      */
int sassd_hash_build(const int32_t *indices, const int32_t *n_ptr, int cap, int D, int H, int W,
                     int batch_size, void *table, size_t table_bytes, int32_t *status, void *stream);
int sassd_rulebook_subm(const int32_t *indices, const int32_t *n_ptr, int cap, int D, int H, int W,
                        int batch_size, const void *table, size_t table_bytes, int32_t *nbr,
                        void *stream);
/* SparseConv3d(k=3, s=2, p=1): out dims (in+2-2-1)/2+1; out rows ascending in linear (b,z,y,x). */
size_t sassd_rulebook_conv_workspace_bytes(int D, int H, int W, int batch_size);
int sassd_rulebook_conv(const int32_t *in_indices, const int32_t *n_in_ptr, int cap_in, int D, int H,
                        int W, int batch_size, const void *in_table, size_t in_table_bytes,
                        int32_t *out_indices, int32_t *n_out_ptr, int cap_out, int32_t *nbr,
                        int32_t *status, void *workspace, size_t workspace_bytes, void *stream);
int sassd_rulebook_pairs(const int32_t *nbr, const int32_t *n_out_ptr, int cap_out, int K,
                         int32_t *pairs /*[K,2,cap_out]*/, int32_t *pair_num /*[K]*/, void *stream);

/* Fused rulebook PYRAMID: all gather tables of a stack of `levels` resolutions (level l: SubMConv3d k=3 rulebook,
 * indice_key "subm<l>"; l-1 -> l: SparseConv3d(k=3,s=2,p=1) rulebook + output coordinates) -- the seven
 * get_indice_pairs calls of VxNet (cmn.py:147-173, 197-206) in one fill + 2 launches per level (9 for VxNet; the
 * per-op chain above takes 30), or in one fill + ONE persistent launch (flags).
 *   indices[l]  [caps[l],4] i32 (b,z,y,x): level 0 is the input, levels >= 1 are written (ascending linear order)
 *   n_ptrs[l]   device int32 row counts: level 0 is the input, levels >= 1 are written
 *   D,H,W       spatial shape of level 0; level l+1 = (dim-1)/2+1 per axis
 *   nbr_subm[l] [caps[l],27] or NULL;  nbr_down[l] (l >= 1) [caps[l],27]: strided table from level l-1 into level l
 *   level_begin/level_end   build levels [begin, end) only (a caller that overlaps the tables with the convolutions
 *               issues one call per level and records an event after each); begin = 0 also resets the workspace.
 *   flags       SASSD_PYRAMID_PERSISTENT: all phases in ONE launch separated by agent-scope grid barriers (needs
 *               level_begin = 0, level_end = levels; bits 8..15 = workgroups per compute unit, 0 = default 2, max 4;
 *               a barrier that times out sets SASSD_ST_GRID_SYNC).  Per call, no process-wide state.
 *   workspace   16-byte aligned (cleared by 16-byte stores).
 * All pointer arrays are HOST arrays of device pointers. */
#define SASSD_PYRAMID_PERSISTENT 1
size_t sassd_rulebook_pyramid_workspace_bytes(int levels, const int *caps, int D, int H, int W, int batch_size);
int sassd_rulebook_pyramid(int levels, int32_t *const *indices, int32_t *const *n_ptrs, const int *caps,
                           int D, int H, int W, int batch_size, int32_t *const *nbr_subm,
                           int32_t *const *nbr_down, int level_begin, int level_end, int flags, int32_t *status,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (a6+a7) Sparse convolution forward with fused per-channel affine (folded eval BatchNorm1d) + ReLU.
 * Replaces spconv `indice_conv_fp32` (+ nn.BatchNorm1d + nn.ReLU, cmn.py:138-173,208-212).
 *   y[o,:] = act( (sum_k x[nbr[o,k],:] @ w[k]) * scale + shift )
 *   w        [K, Cin, Cout] f32 (spconv layout [kz,ky,kx,Cin,Cout] flattened) -> pack once with
 *            sassd_spconv_pack_weight (sassd_spconv_packed_floats floats; an opaque image -- since round 4
 *            [K][Cin/4][Cout][4], four consecutive input channels of one output channel per 16-byte piece).
 *   nbr NULL = identity rulebook with K=1 (the 1x1x1 `extra_conv` shortcut, cmn.py:208-212).
 *   scale/shift may be NULL (1 / 0).  (Cin,Cout) in {(4,16),(16,16),(16,32),(32,32),(32,64),(64,64)}.
 * ---------------------------------------------------------------------------------------------- */
size_t sassd_spconv_packed_floats(int K, int Cin, int Cout);
int sassd_spconv_pack_weight(const float *w, int K, int Cin, int Cout, float *packed, void *stream);
int sassd_spconv_fwd(const float *x, const int32_t *nbr, const int32_t *n_out_ptr, int cap_out,
                     const float *w_packed, int K, int Cin, int Cout, const float *scale,
                     const float *shift, int relu, float *y, int cfg, void *stream);

/* (a15) Sparse convolution backward (training; replaces spconv `indice_conv_backward_fp32`).
 *   sassd_rulebook_transpose   nbrT[i,k] = o  for every rulebook entry nbr[o,k] = i  (-1 elsewhere); nbrT [cap_in,27]
 *   sassd_spconv_pack_weight_t forward weight w [K,Cin,Cout] -> packed image of W[k]^T for the data gradient
 *   sassd_spconv_bwd_data      dx[i,:] = sum_k dy[nbrT[i,k],:] @ W[k]^T       (dx [cap_in,Cin], dy [cap_out,Cout])
 *                              (nbrT NULL + K = 1: the 1x1x1 layer)
 *   sassd_spconv_bwd_weight    dw[k] (+)= sum_{o: nbr[o,k]>=0} x[nbr[o,k],:]^T (x) dy[o,:]   (dw [K,Cin,Cout];
 *                              deterministic two-stage reduction through the caller's workspace)
 * Gradients w.r.t. the fused scale/shift/ReLU epilogue are the caller's (elementwise). */
int sassd_rulebook_transpose(const int32_t *nbr, const int32_t *n_out_ptr, int cap_out, int32_t *nbrT,
                             int cap_in, void *stream);
int sassd_spconv_pack_weight_t(const float *w, int K, int Cin, int Cout, float *packed, void *stream);
int sassd_spconv_bwd_data(const float *dy, const int32_t *nbrT, const int32_t *n_in_ptr, int cap_in,
                          const float *wT_packed, int K, int Cin, int Cout, float *dx, int cfg, void *stream);
size_t sassd_spconv_bwd_weight_workspace_bytes(int cap_out, int K, int Cin, int Cout);
int sassd_spconv_bwd_weight(const float *x, const float *dy, const int32_t *nbr, const int32_t *n_out_ptr,
                            int cap_out, int K, int Cin, int Cout, float *dw, int accumulate, int cfg,
                            void *workspace, size_t workspace_bytes, void *stream);

/* (a8) SparseConvTensor.dense() + view (cmn.py:112-114): out [B, C*D, H, W] f32, zero filled here.
 * channel_order 0: channel = c*D + d (reference); 1: channel = d*C + c (internal, conv0 weights permuted). */
int sassd_densify(const float *feats, const int32_t *indices, const int32_t *n_ptr, int cap, int C,
                  int D, int H, int W, int batch_size, int channel_order, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (a9,a10,a12) Dense 2-D convolution (3x3 pad 1 or 1x1), NCHW fp32, on fp32 MFMA (v_mfma_f32_32x32x2_f32),
 * fused per-channel affine (folded BatchNorm2d / bias) + optional ReLU.  Replaces torch.nn.Conv2d +
 * BatchNorm2d + ReLU in BEVNet (cmn.py:233-282), SSDRotateHead (ssd_rotate_head.py:120-125) and
 * PSWarpHead.convs (:424-429).   w torch layout [Cout,Cin,k,k]; pack once.
 * ---------------------------------------------------------------------------------------------- */
size_t sassd_conv2d_packed_floats(int Cin, int Cout, int ksize);
int sassd_conv2d_pack_weight(const float *w, int Cout, int Cin, int ksize, float *packed, void *stream);
int sassd_conv2d_fwd(const float *x, const float *w_packed, const float *scale, const float *shift,
                     int relu, float *y, int batch, int Cin, int Cout, int H, int W, int ksize,
                     void *stream);
/* ... with a per-call `cfg` word (0 = sassd_conv2d_fwd): ablation switches of tools/run_conv.py -- 1 = no staging DMA after the
 * first chunk, 2 = no chunk barrier. */
int sassd_conv2d_fwd_cfg(const float *x, const float *w_packed, const float *scale, const float *shift, int relu, float *y,
                         int batch, int Cin, int Cout, int H, int W, int ksize, int cfg, void *stream);

/* 1x1 convolution with <= 32 output channels (the fused SSD head, ssd_rotate_head.py:120-125; the second conv of the
 * part-sensitive head, :424-429) as an HBM stream on the vector ALU: same contract as sassd_conv2d_fwd with ksize 1, but
 * the weights are handed over TRANSPOSED and zero-padded, wT [Cin][CO] fp32, CO = sassd_conv1x1_narrow_pad(Cout), 16-byte
 * aligned. */
int sassd_conv1x1_narrow_supported(int Cin, int Cout);
int sassd_conv1x1_narrow_pad(int Cout);
int sassd_conv1x1_narrow_fwd(const float *x, const float *wT, const float *scale, const float *shift, int relu,
                             float *y, int batch, int Cin, int Cout, int H, int W, void *stream);

/* Winograd F(2x2,3x3) variant of the 3x3 / stride 1 / pad 1 convolution (the BEVNet layers cmn.py:240-262): 2.25x
 * fewer multiplications on the fp32 MFMA, transforms fused (nothing but x, packed weights and y touches HBM).  Same
 * epilogue as sassd_conv2d_fwd (y = relu?(conv * scale[co] + shift[co]), NULL scale/shift = 1/0).  Requires
 * Cin % 32 == 0, Cout % 32 == 0, even H, W % 4 == 0 and W >= 64 (sassd_conv2d_wino_supported; anything else takes
 * sassd_conv2d_fwd); relative error ~1e-6 vs the direct kernel. */
int sassd_conv2d_wino_supported(int Cin, int Cout, int H, int W);
size_t sassd_conv2d_wino_packed_floats(int Cin, int Cout);
int sassd_conv2d_wino_pack_weight(const float *w, int Cout, int Cin, float *packed, void *stream);
int sassd_conv2d_wino_fwd(const float *x, const float *w_packed, const float *scale, const float *shift, int relu,
                          float *y, int batch, int Cin, int Cout, int H, int W, void *stream);

/* The same layers through Winograd F(4x4,3x3) as THREE launches: input transform -> 36 GEMMs
 * (Cout x Cin x tiles) -> output transform with the folded BatchNorm / bias / ReLU epilogue.  4x fewer multiplications
 * than the direct convolution (1.78x fewer than F(2x2)); fp32 rounding error ~6x the direct kernel's (2.4e-6 relative after
 * seven layers).  The GEMMs compute their fp32 products on the bf16 MFMA over operands split exactly into three bf16
 * pieces in registers (eight of the nine piece products, fp32 accumulation: the error of the fp32 MFMA at half its
 * cycles); tensors, packed weights and workspace stay fp32.  Needs Cin % 32 == 0, Cout % 256 == 0, H % 4 == 0,
 * W % 4 == 0 and a caller workspace (transformed input + product tensors, 36 planes each). */
int sassd_conv2d_wino4_supported(int Cin, int Cout, int H, int W);
size_t sassd_conv2d_wino4_packed_floats(int Cin, int Cout);
int sassd_conv2d_wino4_pack_weight(const float *w /*[Cout,Cin,3,3]*/, int Cout, int Cin, float *packed, void *stream);
size_t sassd_conv2d_wino4_workspace_bytes(int batch, int Cin, int Cout, int H, int W);
int sassd_conv2d_wino4_fwd(const float *x, const float *w_packed, const float *scale, const float *shift, int relu,
                           float *y, int batch, int Cin, int Cout, int H, int W, int cfg, void *workspace,
                           size_t workspace_bytes, void *stream);

/* Chained 3x3 layers with the activation map between them kept in the transform domain: one call = one layer,
 *   src_products == 0: V = input transform of the NCHW map x (as sassd_conv2d_wino4_fwd);
 *   src_products != 0: V = fused output->input transform of the products the PREVIOUS chain call left in `workspace`
 *                      (that call's Cout = this call's Cin, same batch / H / W / cmax; its folded BatchNorm / ReLU passed
 *                      here as prev_scale / prev_shift / prev_relu) -- the map between the two layers is never written to
 *                      HBM (cmn.py:240-262: conv0 .. conv6 of the BEV stack);
 *   then the 36 GEMMs; y != NULL: output transform + scale / shift / relu into the NCHW map y, y == NULL: the products
 *   stay in the workspace for the next call.  cmax >= every Cin / Cout of the chain fixes the workspace layout
 *   (sassd_conv2d_wino4_chain_workspace_bytes(batch, cmax, H, W)).  sassd_conv2d_wino4_chain_supported: the fused
 *   transform keeps one (H + 2) x (W + 2) plane in LDS (<= 160 KB).  src_products = 1 additionally requires that this
 *   layer and the previous one (whose Cout is this Cin) pad their tile count to the same plane stride -- true whenever
 *   both have the same Cout; SASSD_EINVAL otherwise. */
int sassd_conv2d_wino4_chain_supported(int Cin, int Cout, int H, int W);
size_t sassd_conv2d_wino4_chain_workspace_bytes(int batch, int cmax, int H, int W);
int sassd_conv2d_wino4_chain(const float *x, int src_products, const float *prev_scale, const float *prev_shift,
                             int prev_relu, const float *w_packed, const float *scale, const float *shift, int relu,
                             float *y, int batch, int Cin, int Cout, int cmax, int H, int W, const int32_t *tile_map,
                             const int32_t *prev_tile_map, int cfg, void *workspace, size_t workspace_bytes,
                             void *stream);
/* A NARROW layer (<= 64 output channels) at the end of a chain: the part-sensitive head's 3x3 conv 256 -> 28
 * (ssd_rotate_head.py:424-429) on conv6's products.  One call = the fused transform of the previous call's products (prev_* = that
 * layer's folded BatchNorm / ReLU; y_prev, optional: its NCHW activation map, stored from the LDS plane for its other readers --
 * BEVNet conv7, cmn.py:262), the 36 GEMMs on a 64-channel block, the output transform + scale / shift / relu into y [B,Cout,H,W].
 * Weights: sassd_conv2d_wino4_pack_weight_narrow ([36][Cin][64], zero padded).  Default GEMM geometry only. */
size_t sassd_conv2d_wino4_narrow_packed_floats(int Cin);
int sassd_conv2d_wino4_pack_weight_narrow(const float *w /*[Cout,Cin,3,3]*/, int Cout, int Cin, float *packed, void *stream);
int sassd_conv2d_wino4_chain_tail(const float *prev_scale, const float *prev_shift, int prev_relu, float *y_prev,
                                  const float *w_packed64, const float *scale, const float *shift, int relu, float *y,
                                  int batch, int Cin, int Cout, int cmax, int H, int W, const int32_t *prev_tile_map, int cfg,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* Active-tile map of a SPARSE input map (BEVNet conv0 reads SparseConvTensor.dense(), cmn.py:112-114,240: 56 % of the 4x4
 * tiles of a KITTI frame have an occupied pixel in their 6x6 patch).  indices [cap,4] (b, z, y, x) = the sparse rows that
 * were densified.  A chain call with `tile_map` (src_products == 0) transforms and multiplies the active tiles only
 * (compacted columns); the NEXT chain call passes the same map as `prev_tile_map` (or this call, with y != NULL, its own):
 * an inactive tile's products are exactly zero, so the result is bit-identical to the dense launch.  NULL = all tiles.
 * tile_map: sassd_wino4_tile_map_ints(batch, H, W) int32 (0 = unsupported: more than 65536 tiles). */
size_t sassd_wino4_tile_map_ints(int batch, int H, int W);
int sassd_wino4_tile_map(const int32_t *indices, const int32_t *n_ptr, int cap, int batch, int H, int W,
                         int32_t *tile_map, void *stream);

/* 1x1 convolution with >= 128 output channels (BEVNet conv7, cmn.py:262) as a plain fp32-MFMA GEMM over the NCHW
 * tensor (y[b] [Cout x HW] = W [Cout x Cin] . x[b] [Cin x HW]) with the folded BatchNorm / bias / ReLU epilogue; the
 * kernel of the Winograd F(4x4) products with one problem per image.  Needs Cin % 32 == 0, Cout % 128 == 0 and H*W
 * divisible by one of 64 / 96 / 128 / 160 / 192; weights packed [Cin][Cout] (Cin*Cout floats). */
int sassd_conv1x1_gemm_supported(int Cin, int Cout, int H, int W);
int sassd_conv1x1_gemm_pack_weight(const float *w /*[Cout,Cin,1,1]*/, int Cout, int Cin, float *packed, void *stream);
int sassd_conv1x1_gemm_fwd(const float *x, const float *w_packed, const float *scale, const float *shift, int relu,
                           float *y, int batch, int Cin, int Cout, int H, int W, int cfg, void *stream);

/* BASELINE configs[2] (bf16 training): the 3x3 pad-1 BEV convolutions (cmn.py:240-262) with bf16 MFMA operands --
 * direct implicit GEMM on v_mfma_f32_32x32x16_bf16, fp32 accumulation, NCHW fp32 activations in and out, optional
 * per-channel bias.  Weights are packed once per update ([Cout,Cin,3,3] fp32 -> bf16 [tap][Cin32/8][Cout][8],
 * sassd_conv2d_bf16_packed_elems 16-bit elements).  The data gradient is the same call on dy with the weights
 * transposed and the taps mirrored.  Supported: Cout % 32 == 0, W >= 16 and W % 4 == 0 (the last 16-column tile may be partial: the 188-wide Waymo-scale map; else SASSD_EINVAL; the caller keeps the
 * fp32 kernels); Cin is padded to a multiple of 32 with zero weights inside the pack. */
int sassd_conv2d_bf16_supported(int Cin, int Cout, int H, int W);
size_t sassd_conv2d_bf16_packed_elems(int Cin, int Cout);
int sassd_conv2d_bf16_pack_weight(const float *w, int Cout, int Cin, void *packed, void *stream);
int sassd_conv2d_bf16_fwd(const float *x, const void *w_packed, const float *shift, float *y, int batch, int Cin,
                          int Cout, int H, int W, void *stream);
/* ... with a per-call `cfg` word (0 = sassd_conv2d_bf16_fwd; tools / tests only): low byte = compile-time ablation variant of the
 * kernel (tools/run_bf16_conv.py), bits 8-15 = forced workgroup count (long runs of tiles), 0x10000 / 0x20000 / 0x40000 =
 * wave-priority and loader-order A/B switches, 0x80000 = Cout = 320 on 128-cout instead of 160-cout workgroups. */
int sassd_conv2d_bf16_fwd_cfg(const float *x, const void *w_packed, const float *shift, float *y, int batch, int Cin,
                              int Cout, int H, int W, int cfg, void *stream);
/* The same convolution over relu(batchnorm(x)), the normalisation applied by the kernel's loader waves (round 6): x is the RAW
 * output of the previous convolution, in_affine [3][Cin] = mean | invstd * gamma | beta as sassd_bn2d_stats writes it.  The
 * operand that reaches the MFMA is bit-identical to the one sassd_bn2d_relu_fwd + sassd_conv2d_bf16_fwd produce, and the
 * normalised map (cmn.py:236-237 between two BEV layers) is never written to HBM.  Cin <= 1024. */
int sassd_conv2d_bf16_bnrelu_fwd(const float *x, const float *in_affine, const void *w_packed, const float *shift, float *y,
                                 int batch, int Cin, int Cout, int H, int W, void *stream);

/* 1x1 convolution on the bf16 MFMA (round 6; same arithmetic contract as sassd_conv2d_bf16_fwd: weights rounded to bf16 at pack
 * time, activations on their way into the MFMA, fp32 accumulation, fp32 NCHW tensors in and out): BEVNet's conv7 (cmn.py:262),
 * its data gradient, and the data gradient of the SSD head's 1x1 convs (ssd_rotate_head.py:120-125) under
 * set_bev_precision("bf16") -- HBM streams of 144 MB that took 150 us each on the fp32-MFMA kernel.  x [B,Cin,HW] -> y
 * [B,Cout,HW] (+ shift[Cout], may be NULL).  Cin <= 256 (padded inside the pack), any Cout (64-channel groups, the last
 * one masked), HW % 4 == 0; x 8-byte, y and the pack 16-byte aligned.  pack_weight: w fp32 [Cout][Cin], or -- transposed = 1 --
 * the [Cin][Cout] array of the forward layer read as its transpose (the data gradient's weights, no host-side transpose);
 * sassd_conv1x1_bf16_packed_elems 16-bit elements. */
int sassd_conv1x1_bf16_supported(int Cin, int Cout, int HW);
size_t sassd_conv1x1_bf16_packed_elems(int Cin, int Cout);
int sassd_conv1x1_bf16_pack_weight(const float *w, int Cout, int Cin, int transposed, void *packed, void *stream);
int sassd_conv1x1_bf16_fwd(const float *x, const void *w_packed, const float *shift, float *y, int batch, int Cin, int Cout,
                           int HW, void *stream);

/* Training: weight gradient of the same convolutions (autograd of nn.Conv2d at cmn.py:240-262 and
 * ssd_rotate_head.py:120-125,424-429; cuDNN in the reference).  x [B,Cin,H,W], dy [B,Cout,H,W] NCHW fp32 ->
 * dw [Cout,Cin,k,k] (torch layout), overwritten or accumulated.  Split-K over pixels with a deterministic second-stage
 * reduction; workspace from sassd_conv2d_wgrad_workspace_bytes.  The data gradient is sassd_conv2d_fwd on dy with the
 * weights transposed and the taps mirrored (sassd.autograd.Conv2dFn). */
size_t sassd_conv2d_wgrad_workspace_bytes(int batch, int Cin, int Cout, int H, int W, int ksize);
int sassd_conv2d_bwd_weight(const float *x, const float *dy, float *dw, int batch, int Cin, int Cout, int H, int W,
                            int ksize, int accumulate, void *workspace, size_t workspace_bytes, void *stream);
/* BASELINE configs[2] (bf16 training): the same contraction with x and dy rounded to bf16 (round-to-nearest-even) on
 * their way into LDS, multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; dw stays fp32.  W must be even
 * (SASSD_EINVAL otherwise).  What the reference gets from cuDNN under torch autocast / apex O1. */
int sassd_conv2d_bwd_weight_bf16(const float *x, const float *dy, float *dw, int batch, int Cin, int Cout, int H, int W,
                                 int ksize, int accumulate, void *workspace, size_t workspace_bytes, void *stream);
/* ... with the X operand = relu(batchnorm(x)) applied on the way into LDS (x_affine as above; 3x3 only): the weight gradient of
 * a layer whose input map was never materialised (sassd_conv2d_bf16_bnrelu_fwd). */
int sassd_conv2d_bwd_weight_bf16_bnrelu(const float *x, const float *x_affine, const float *dy, float *dw, int batch, int Cin,
                                        int Cout, int H, int W, int ksize, int accumulate, void *workspace,
                                        size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (f-3 / a18) fused training targets and RPN loss.
 *
 * sassd_assign_targets: mmdet/core/bbox3d/target_ops.py:139-277 (create_target_np) for a whole batch in two launches.
 *   anchors [A,7] (anchors_per_sample = 0) or [B,A,7]; anchor_mask [B,A] u8 or NULL; ground truths of all samples
 *   concatenated: gt_boxes [T,7], gt_classes [T] i64 or NULL (all 1), gt_ok [T] u8 or NULL, gt_offsets [B+1] i32
 *   (device).  Similarity = NearestIouSimilarity (iou3d_utils.py:163-183) computed in the kernel, or -- overlaps != NULL
 *   -- a precomputed row-major [A, G_b] matrix per sample starting at overlaps + overlap_offsets[b] (i64, device).
 *   Outputs: labels i64 (-1 ignore / 0 negative / class), targets (second_box_encode, ssd_rotate_head.py:15-50, zero
 *   for non-positives), best_overlap (or NULL), each with out_sample_stride anchors between samples (>= A: lets
 *   several classes write interleaved [B, classes, A] tensors); num_pos[b] += positives (zeroed first when
 *   zero_num_pos).  Workspace: sassd_assign_targets_workspace_bytes.
 * sassd_rpn_loss: ssd_rotate_head.py:128-314 loss() in one pass: loss_sums[3] = (smooth-L1 with sin-difference, sigmoid
 *   focal, direction cross-entropy) with NormByNumPositives weights, and the gradients of those three sums with
 *   respect to box_preds [B,A,7], cls_preds [B,A,num_class], dir_preds [B,A,2] (dir_preds may be NULL). */
size_t sassd_assign_targets_workspace_bytes(int batch, int n_anchors, int total_gt);
int sassd_assign_targets(const float *anchors, int anchors_per_sample, const uint8_t *anchor_mask, int n_anchors,
                         int batch, const float *gt_boxes, const int64_t *gt_classes, const uint8_t *gt_ok,
                         const int32_t *gt_offsets, int total_gt, const float *overlaps, const int64_t *overlap_offsets,
                         float matched_threshold, float unmatched_threshold, int64_t *labels, float *targets,
                         float *best_overlap, size_t out_sample_stride, int32_t *num_pos, int zero_num_pos,
                         void *workspace, size_t workspace_bytes, void *stream);
/* Guided-anchor selection for training (ssd_rotate_head.py:316-388) without a host round trip: ascending indices of the
 * anchors with anchor_mask set (NULL = all) and sigmoid(max_c cls_preds[b,a,c]) > score_thr go to sel[b][0..counts[b]) of
 * a fixed-capacity [B,cap] int64 buffer (remaining entries p -> p: valid, distinct indices); *overflow is set (never cleared) when a sample had more
 * than cap (the surplus is dropped).  Workspace: sassd_guided_select_workspace_bytes. */
size_t sassd_guided_select_workspace_bytes(int batch, int n_anchors);
int sassd_guided_select(const float *cls_preds, const uint8_t *anchor_mask, int n_anchors, int batch, int num_class,
                        float score_thr, int cap, int64_t *sel, int32_t *counts, int32_t *overflow, void *workspace,
                        size_t workspace_bytes, void *stream);
size_t sassd_rpn_loss_workspace_bytes(int batch, int n_anchors);
int sassd_rpn_loss(const float *box_preds, const float *cls_preds, const float *dir_preds, int num_class,
                   const int64_t *labels, const float *targets, const float *anchors, int anchors_per_sample,
                   const int32_t *num_pos, int n_anchors, int batch, float *grad_box, float *grad_cls, float *grad_dir,
                   float *loss_sums, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (f-1) anchors_mask: mmdet/datasets/kitti.py:333-343 with geometry.py:676-710
 * (sparse_sum_for_anchors_mask -> cumsum -> fused_get_anchors_area > area_threshold).
 *   coors [n,4] (b,z,y,x) rows of ONE batch element are selected by batch_idx
 *   anchors_bv [A,4] f32 (xmin,ymin,xmax,ymax); mask [A] u8
 * ---------------------------------------------------------------------------------------------- */
size_t sassd_anchor_mask_workspace_bytes(int H0, int W0);
int sassd_anchor_mask(const int32_t *coors, const int32_t *row_begin_ptr, const int32_t *row_end_ptr,
                      int H0, int W0, const float *anchors_bv, int n_anchors, const float *voxel_size,
                      const float *coors_range, float area_threshold, uint8_t *mask, void *workspace,
                      size_t workspace_bytes, void *stream);
/* The same for `batch` samples in one launch sequence: sample b owns the coordinate rows [row_offsets[b],
 * row_offsets[b+1]) (device int32[batch+1], what sassd_voxelize's row_offset chain leaves behind), mask[b*n_anchors..]
 * and one sassd_anchor_mask_workspace_bytes slice of the workspace (batch slices in total). */
int sassd_anchor_mask_batch(const int32_t *coors, const int32_t *row_offsets, int batch, int H0, int W0,
                            const float *anchors_bv, int n_anchors, const float *voxel_size,
                            const float *coors_range, float area_threshold, uint8_t *mask, void *workspace,
                            size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (a11) get_guided_anchors, test path (ssd_rotate_head.py:307-372) + second_box_decode (:53-91).
 *   box/cls/dir: raw conv outputs [B, ncls*A*7 | ncls*A*ncls | ncls*A*2, H, W] (NCHW), channel strides
 *   given in floats via *_batch_stride (lets the three heads live in one fused conv output).
 *   anchors [Atot,7] (Atot = ncls*H*W*A), mask [B,Atot] u8.  Keeps anchors with mask && max-class
 *   sigmoid score > thr, in ascending anchor order; applies the direction flip.
 *   out: guided [B,capK,7], labels [B,capK] i32, scores [B,capK] f32 (rpn score, extra), counts [B] i32.
 * ---------------------------------------------------------------------------------------------- */
size_t sassd_decode_filter_workspace_bytes(int batch, int n_anchors_total);
int sassd_decode_filter(const float *box, const float *cls, const float *dir, size_t batch_stride,
                        int batch, int num_class, int anchors_per_loc, int H, int W,
                        const float *anchors, const uint8_t *mask, float thr, float *guided,
                        int32_t *labels, float *scores, int32_t *counts, int capK, int32_t *status,
                        void *workspace, size_t workspace_bytes, void *stream);

/* (a12) PSWarpHead sampling (ssd_rotate_head.py:374-414,438-442): feat [B,parts,H,W] (parts = 4*7),
 * guided [B,capK,7], counts [B] -> logits [B,capK] = mean_k bilinear(feat[k], grid point k). */
int sassd_pswarp_sample(const float *feat, int batch, int H, int W, const float *guided,
                        const int32_t *counts, int capK, float grid_off_x, float grid_off_y,
                        float spatial_scale, float *logits, void *stream);

/* backward of sassd_pswarp_sample (training, PSWarpHead.loss): dlogits [B,capK] -> ACCUMULATES into dfeat [B,28,H,W]
 * (caller zeroes; float atomics) and writes dguided [B,capK,7] (d/d x,y,w,l,r; z,h columns zero; may be NULL) --
 * torch.nn.functional.grid_sample differentiates w.r.t. both input and grid (ssd_rotate_head.py:400-414). */
int sassd_pswarp_sample_bwd(const float *feat, int batch, int H, int W, const float *guided,
                            const int32_t *counts, int capK, float grid_off_x, float grid_off_y,
                            float spatial_scale, const float *dlogits, float *dfeat, float *dguided,
                            void *stream);

/* (a13+a14) get_rescore_bboxes (ssd_rotate_head.py:487-533): sigmoid, > score_thr, BEV boxes
 * (iou3d_utils.py:47-60), stable descending sort, rotated NMS (iou3d_kernel.cu:250-292 + iou3d.cpp:100-116).
 *   out_boxes [B,capD,7], out_scores [B,capD], out_labels [B,capD] i32, out_counts [B] i32. */
size_t sassd_rescore_nms_workspace_bytes(int batch, int capK);
int sassd_rescore_nms(const float *guided, const float *logits, const int32_t *labels,
                      const int32_t *counts, int batch, int capK, float score_thr, float iou_thr,
                      float *out_boxes, float *out_scores, int32_t *out_labels, int32_t *out_counts,
                      int capD, int32_t *status, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * iou3d_cuda operator surface (mmdet/ops/iou3d/src/iou3d.cpp:31,52,73; kernels iou3d_kernel.cu:223-292).
 * boxes (x1,y1,x2,y2,ry) f32.  nms: boxes sorted by descending score; keep [n] i64 and num_keep are DEVICE
 * buffers (the reference fills a CPU LongTensor after a blocking D2H of the mask, iou3d.cpp:92-94).
 * ---------------------------------------------------------------------------------------------- */
int sassd_boxes_overlap_bev(const float *boxes_a, int num_a, const float *boxes_b, int num_b,
                            float *ans_overlap, void *stream);
int sassd_boxes_iou_bev(const float *boxes_a, int num_a, const float *boxes_b, int num_b,
                        float *ans_iou, void *stream);
size_t sassd_nms_workspace_bytes(int n);
int sassd_nms_gpu(const float *boxes, int n, float thresh, int64_t *keep, int32_t *num_keep,
                  void *workspace, size_t workspace_bytes, void *stream);
/* iou3d_cuda.nms_normal_gpu (iou3d.cpp:123-172, iou3d_kernel.cu:295-348): the same greedy suppression over the
 * axis-aligned IoU of the (x1,y1,x2,y2) rectangles, rotation ignored.  Same arguments / workspace as sassd_nms_gpu. */
int sassd_nms_normal_gpu(const float *boxes_sorted, int n, float thresh, int64_t *keep, int32_t *num_keep,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (a16, a17) training-side point operators of the auxiliary network.
 * pointnet2_cuda (mmdet/ops/pointnet2/src/interpolate.cpp:13,25,40; kernels interpolate_gpu.cu:9-146):
 *   three_nn            unknown [N,4] (b,x,y,z), known [M,4] -> dist2 [N,3] f32 (squared), idx [N,3] i32: the three
 *                       nearest known points of the same batch index, first index wins ties
 *   three_interpolate   points [M,C], idx, weight [N,3] -> out [N,C]
 *   three_interpolate_grad  grad_out [N,C] -> ACCUMULATES into grad_points [M,C] (caller zeroes it; float atomics,
 *                       like the reference interpolate_gpu.cu:143-145)
 * points_op_cpu (mmdet/ops/points_op/src/points_op.cpp:107-144):
 *   pts_in_boxes3d      pts [N,3], boxes3d [M,7] -> pts_flag [M,N] i32 (fully written), reg_target [N,3] f32
 *                       (only rows of points inside a box are written: caller zeroes, as points_op/__init__.py:8-9)
 * ---------------------------------------------------------------------------------------------- */
int sassd_three_nn(int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx,
                   void *stream);
/* Exact accelerated variant: the known points are counting-sorted into a uniform BEV grid (batch, y, x) of
 * `cell`-sized columns starting at (x0, y0) with nx * ny cells per batch element (coordinates outside are clamped
 * into the border cells), and each query scans rings of cells until its 3rd best distance is provably final.
 * Results are bit-identical to sassd_three_nn for ANY inputs (same ranking by (distance,row), same fp32 distance
 * arithmetic); cost ~O((N+M) * points-per-neighbourhood) instead of O(N*M).  Batch indices must be 0..batch_size-1;
 * nx*ny*batch_size <= 2^20. */
size_t sassd_three_nn_binned_workspace_bytes(int m, int nx, int ny, int batch_size);
int sassd_three_nn_binned(int n, int m, const float *unknown, const float *known, float x0, float y0, float cell,
                          int nx, int ny, int batch_size, float *dist2, int32_t *idx, void *workspace,
                          size_t workspace_bytes, void *stream);
int sassd_three_interpolate(int c, int m, int n, const float *points, const int32_t *idx, const float *weight,
                            float *out, void *stream);
int sassd_three_interpolate_grad(int c, int n, int m, const float *grad_out, const int32_t *idx,
                                 const float *weight, float *grad_points, void *stream);
int sassd_pts_in_boxes3d(const float *pts, int n, const float *boxes3d, int m, int32_t *pts_flag,
                         float *reg_target, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (f-2) KITTI-evaluation rotated IoU.  Replaces `rotate_iou_gpu_eval(boxes, query_boxes, criterion, device_id)` of
 * mmdet/core/post_processing/rotate_nms_gpu.py:594-627 (numba.cuda kernel :548-591, device functions :153-388), called
 * by mmdet/core/evaluation/kitti_eval.py for the BEV / 3-D overlap matrices.
 *   boxes [n,5], query_boxes [k,5] f32 device: (cx, cy, x_dim, y_dim, angle);  iou [n,k] f32 device
 *   iou[i][j] = inter / (area_q + area_b - inter)  (criterion -1),  inter / area(query j) (0),  inter / area(box i) (1),
 *               inter (anything else) -- the reference's argument order devRotateIoUEval(query, box).
 * ---------------------------------------------------------------------------------------------- */
int sassd_rotate_iou_eval(const float *boxes, int n, const float *query_boxes, int k, int criterion, float *iou,
                          void *stream);

/* (f-2) KITTI-evaluation matching statistics -- HOST function (no device memory, no stream): the greedy
 * ground-truth <-> detection assignment is sequential per image and negligible next to the overlap matrices above.
 * Replaces `compute_statistics_jit` (mmdet/core/evaluation/kitti_eval.py:165-283, numba CPU) and the loop over images x
 * score thresholds `fused_compute_statistics` (:296-343) for ONE part of the image list (eval_class_v3 :598-648).
 *   overlaps [sum dt, ld] f64 host, row-major: overlap of detection j with ground truth i; image p owns the block
 *            rows [sum_{q<p} dt_nums[q], +dt_nums[p]) x columns [sum_{q<p} gt_nums[q], +gt_nums[p]);  ld >= sum gt
 *   gt_datas [sum gt,5] (bbox x1 y1 x2 y2, alpha)   dt_datas [sum dt,6] (bbox, alpha, score)   dontcares [sum dc,4]
 *   ignored_gts [sum gt] / ignored_dets [sum dt] int64: 0 counted, 1 neutral, -1 other class (clean_data :39-93)
 *   metric 0 image bbox (detections inside DontCare boxes are not false positives), 1 BEV, 2 3-D
 *   n_thr == 0: first pass (compute_fp=False, thresh 0): tp_scores[<= sum gt] receives the score of every matched
 *               detection in image order, *n_tp_scores their count;  pr (optional, [4]) += (tp, 0, fn, 0)
 *   n_thr  > 0: pr [n_thr,4] f64 += (tp, fp, fn, sum over true positives of (1+cos(alpha_gt-alpha_dt))/2 when
 *               compute_aos) per threshold; detections scoring under thresholds[t] are left out
 * Returns SASSD_EINVAL for negative counts, metric outside 0..2, ld < sum gt or a missing output array. */
int sassd_kitti_eval_statistics(const double *overlaps, int64_t ld, int n_img, const int64_t *gt_nums,
                                const int64_t *dt_nums, const int64_t *dc_nums, const double *gt_datas,
                                const double *dt_datas, const double *dontcares, const int64_t *ignored_gts,
                                const int64_t *ignored_dets, int metric, double min_overlap, const double *thresholds,
                                int n_thr, int compute_aos, double *pr, double *tp_scores, int64_t *n_tp_scores);

/* ------------------------------------------------------------------------------------------------
 * (f-4) Training-side augmentation and offline data preparation.  The reference runs these as numba CPU loops inside
 * DataLoader workers (mmdet/core/point_cloud/point_augmentor.py, mmdet/core/bbox3d/geometry.py, tools/create_data.py).
 * Device entry points work in place on a point cloud that is already in HBM ([n, stride] f32 rows, xyz first).
 * ---------------------------------------------------------------------------------------------- */

/* `points_in_convex_polygon_3d_jit(points, surfaces)` geometry.py:189-227, the core of `points_in_rbbox` (:63-74, called
 * from kitti.py:222, create_data.py:43,221) and of `remove_outside_points` (:50-61, camera-frustum reduction).
 *   planes [m,6,4] f64 device: (nx, ny, nz, d) of each polytope's 6 surfaces, as `surface_equ_3d_jit` (:176-186) gives
 *   them -- the caller builds them from the box / frustum corners with the reference's own numpy arithmetic;
 *   f32_math != 0: the plane values are float32 numbers (float32 boxes) and the sign is evaluated in float32, as the
 *   reference's dtype rules make it;  mask [n,m] u8 device: 1 when no surface gives n.p + d >= 0. */
int sassd_points_in_polytopes(const float *points, int n, int stride, const double *planes, int m, int f32_math,
                              uint8_t *mask, void *stream);

/* `points_transform_(points, centers, point_masks, loc_transform, rot_transform, valid_mask)` point_augmentor.py:44-62:
 * every point inside a valid box is rotated about that box's centre and shifted with it (first such box wins).
 *   mask [n,m] u8, valid [m] u8, centers [m,3] f32, rot_sin / rot_cos [m] f32 (float64 sine / cosine of the accepted yaw
 *   noise rounded to float32, the entries of the reference's float32 rotation matrix), loc [m,3] f64 -- all device. */
int sassd_points_transform(float *points, int n, int stride, const uint8_t *mask, int m, const uint8_t *valid,
                           const float *centers, const float *rot_sin, const float *rot_cos, const double *loc,
                           void *stream);

/* `random_flip` + `global_rotation` + `global_scaling` on the points, one pass (point_augmentor.py:279-303):
 * y -> -y if flip; [x y z] @ [[c,-s,0],[s,c,0],[0,0,1]]; xyz *= scale. */
int sassd_points_global_transform(float *points, int n, int stride, int flip, float rot_sin, float rot_cos, float scale,
                                  void *stream);

/* The points of the sampled ground-truth objects (`PointAugmentor.sample_all` point_augmentor.py:232-242: np.fromfile
 * per object, += box centre, -= road-plane correction) gathered from a database that is RESIDENT IN HBM.
 *   db_points [P,4] f32;  src_start [n_obj] i64 first database row of object k;  out_start [n_obj+1] i64 prefix sums of
 *   the objects' point counts (out_start[n_obj] == n_out);  shift [n_obj,3] f64;  lower [n_obj] f64 or NULL;
 *   out [n_out,4] f32 -- all device. */
int sassd_paste_objects(const float *db_points, const int64_t *src_start, const int64_t *out_start, int n_obj,
                        int64_t n_out, const double *shift, const double *lower, float *out, void *stream);

/* HOST functions (no device memory, no stream): the sequential O(boxes^2 x tries) decisions.
 * `box_collision_test(boxes, qboxes)` geometry.py:593-672: boxes [n,4,2], qboxes [k,4,2] corner arrays (float64 when
 * is_f64, else float32) -> out [n,k] u8; edge crossings and full containment both count (numba evaluates the
 * reference's `ret[i, j] is False` as an equality test).
 * `noise_per_box(boxes, valid_mask, loc_noises, rot_noises)` point_augmentor.py:73-105: boxes [n,5] f32 (x, y, w, l, yaw),
 * valid [n] u8, loc_noises [n,num_try,3] f64, rot_noises [n,num_try] f64 -> success [n] i64: index of the first draw
 * that keeps box i clear of all others (accepted draws move the box for the later tests), -1 if none / not valid. */
int sassd_box_collision_test(const void *boxes, int n, const void *qboxes, int k, int is_f64, uint8_t *out);
int sassd_noise_per_box(const float *boxes, const uint8_t *valid, const double *loc_noises, const double *rot_noises,
                        int n, int num_try, int64_t *success);

/* ---- training: parameter update ------------------------------------------------------------------------------------
 * Replaces tools/train_utils/__init__.py:57-61 (clip_grad_norm_ + optimizer.step) for optimizer type 'adam_onecycle'
 * (tools/train_utils/optimization/__init__.py:17-30, fastai_optim.py:132-148): decoupled weight decay on every
 * parameter, then Adam(betas=(mom, 0.99), eps, weight_decay=0) -- over one flat fp32 buffer.
 * sassd_grad_sumsq: out[0] = sum(grad^2) (device float, zeroed by the call).
 * sassd_adam_step:  g' = grad * grad_scale * min(1, max_norm / (sqrt(sumsq) * grad_scale + 1e-6)) when grad_sumsq is
 *                   non-NULL and max_norm > 0 (torch clip_grad_norm_ semantics), else grad * grad_scale;
 *                   p *= 1 - wd*lr; m,v Adam moments; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
 *                   `step` is t >= 1.  All pointers 16-byte aligned. */
/* Training-mode BatchNorm1d + ReLU of the sparse blocks (cmn.py:147-173: SubMConv3d / SparseConv3d -> BatchNorm1d(eps
 * 1e-3, momentum 0.01) -> ReLU) over features x [n, C] row-major, two launches each way instead of torch's six.
 *   fwd: y = relu((x - mean) * invstd * gamma + beta) with batch statistics (biased variance); save_mean / save_invstd [C]
 *        are kept for the backward pass; running_mean / running_var (both or neither) are updated like torch (momentum,
 *        unbiased variance).
 *   bwd: dx, dgamma, dbeta from dy (gradient with respect to y) and the saved statistics.
 * C % 4 == 0, 256 % C == 0.  Deterministic: block partials in double, reduced in a fixed order by every block of the
 * second launch (no inter-workgroup hand-off inside a launch).  Workspace: sassd_bn_relu_workspace_bytes(C). */
size_t sassd_bn_relu_workspace_bytes(int C);
int sassd_bn_relu_fwd(const float *x, int n, int C, const float *gamma, const float *beta, float *running_mean,
                      float *running_var, float momentum, float eps, float *y, float *save_mean, float *save_invstd,
                      void *workspace, size_t workspace_bytes, void *stream);
int sassd_bn_relu_bwd(const float *x, const float *dy, int n, int C, const float *gamma, const float *beta,
                      const float *save_mean, const float *save_invstd, float *dx, float *dgamma, float *dbeta,
                      void *workspace, size_t workspace_bytes, void *stream);

/* Training-mode BatchNorm2d + ReLU over NCHW maps x [B, C, H*W] (the BEV stack cmn.py:233-282 and the part-sensitive head
 * ssd_rotate_head.py:424-429: Conv2d -> BatchNorm2d(eps 1e-3, momentum 0.01) -> ReLU), replacing torch's MIOpen BatchNorm +
 * clamp / threshold_backward + MIOpen backward (four passes over the map each way) by two launches each way; same
 * contract as sassd_bn_relu_* with the channel as the SECOND dimension.  HW % 4 == 0, 16-byte aligned tensors.
 * Deterministic (double partials per (channel, split), added in a fixed order by every block of the second launch).
 * Workspace: sassd_bn2d_relu_workspace_bytes(C). */
size_t sassd_bn2d_relu_workspace_bytes(int C);
int sassd_bn2d_relu_fwd(const float *x, int B, int C, int HW, const float *gamma, const float *beta, float *running_mean,
                        float *running_var, float momentum, float eps, float *y, float *save_mean, float *save_invstd,
                        void *workspace, size_t workspace_bytes, void *stream);
/* Statistics only: mean / invstd / running statistics exactly as sassd_bn2d_relu_fwd computes them, plus the affine triple
 * [3][C] = mean | invstd * gamma | beta for consumers that apply BatchNorm + ReLU themselves (sassd_conv2d_bf16_bnrelu_fwd,
 * sassd_conv2d_bwd_weight_bf16_bnrelu); the backward is sassd_bn2d_relu_bwd on the raw map. */
int sassd_bn2d_stats(const float *x, int B, int C, int HW, const float *gamma, const float *beta, float *running_mean,
                     float *running_var, float momentum, float eps, float *save_mean, float *save_invstd, float *affine,
                     void *workspace, size_t workspace_bytes, void *stream);
int sassd_bn2d_relu_bwd(const float *x, const float *dy, int B, int C, int HW, const float *gamma, const float *beta,
                        const float *save_mean, const float *save_invstd, float *dx, float *dgamma, float *dbeta,
                        void *workspace, size_t workspace_bytes, void *stream);

/* The guided-anchor / rescoring tail of the training step on padded tensors with device counts (train_heads.hip):
 *   sassd_guided_decode_fwd  ssd_rotate_head.py:316-388 (train mode): guided[b] = [ground truth of sample b (gt_boxes rows
 *                            gt_off[b] .. gt_off[b+1]); decoded + direction-flipped boxes of the anchors sel[b][0 .. sel_count[b])
 *                            (second_box_decode :53-91, flip :352-356); zeros] as [B, gmax + cap, 7]; counts[b] = G_b +
 *                            min(sel_count[b], cap).  anchors [A,7] (anchors_per_sample 0) or [B,A,7] (1); dir_preds may be NULL.
 *   sassd_guided_decode_bwd  dbox [B,A,7] (zeroed here) <- dguided through the decode (selected anchors only).
 *   sassd_boxes_iou3d_batch  iou3d_utils.py:79-111 boxes_iou3d_gpu of boxes [B, rows, 7] (rows < counts[b] valid, the rest
 *                            give 0) against each sample's ground truth; sample b's [rows, G_b] matrix starts at element
 *                            ov_off[b] of `overlaps` (the layout sassd_assign_targets takes as `overlaps`).
 *   sassd_focal_loss         losses.py:35-62 sigmoid focal loss (gamma 2, alpha .25) of logits [n] against labels [n]
 *                            (-1 ignore / 0 / > 0), weight 1 / max(sum(num_pos[0..nb)), 1): loss_sum[0] and grad [n]. */
int sassd_guided_decode_fwd(const float *box_preds, const float *dir_preds, const float *anchors, int anchors_per_sample,
                            const int64_t *sel, const int32_t *sel_count, const float *gt_boxes, const int32_t *gt_off,
                            int A, int B, int cap, int gmax, float *guided, int32_t *counts, void *stream);
int sassd_guided_decode_bwd(const float *box_preds, const float *anchors, int anchors_per_sample, const int64_t *sel,
                            const int32_t *sel_count, const int32_t *gt_off, int A, int B, int cap, int gmax,
                            const float *dguided, float *dbox, void *stream);
int sassd_boxes_iou3d_batch(const float *boxes, const int32_t *counts, int B, int rows, const float *gt_boxes,
                            const int32_t *gt_off, int gmax, const int64_t *ov_off, float *overlaps, void *stream);
size_t sassd_focal_loss_workspace_bytes(int n);
int sassd_focal_loss(const float *logits, const int64_t *labels, int n, const int32_t *num_pos, int nb, float *loss_sum,
                     float *grad, void *workspace, size_t workspace_bytes, void *stream);

/* The auxiliary point-wise head of SpMiddleFHD in training, fused (cmn.py:27-29 point_fc / point_cls / point_reg, :45-72
 * build_aux_target, :74-104 aux_loss, :121-135,175-189 nearest_neighbor_interpolate, transforms.py:218-223 tensor2points):
 *   sassd_aux_prepare   points [N,4] = (b, voxel mean xyz); voxel centres known[s] [M_s,4] of the three middle tensors
 *                       (indices [M_s,4] (b,z,y,x); centre = idx * vs_s + offset + vs_s / 2 with vs_s = 2, 4, 8 x voxel_size,
 *                       evaluated in the reference's fp32 order); per-point label (inside any ground-truth box of the
 *                       point's own sample: points_op.cpp:92-144 semantics, gt_boxes [T,7] + gt_off [B+1]) and centre
 *                       offsets of the LAST containing box; npos[0] = number of positive points.
 *   (the three 3-NN searches between points and known[s] stay sassd_three_nn_binned calls)
 *   sassd_aux_head_fwd  per point: inverse-distance weights from nn_d2[s] [N,3], interpolation of feats[s] [M_s, C_s]
 *                       (C = 32, 64, 64) at nn_idx[s] [N,3], h = f W1^T (W1 [64,160]), out = h W2^T (W2 [4,64]: point_cls
 *                       row, then the three point_reg rows); loss_sums[0] = sum of the sigmoid focal terms / max(npos, 1),
 *                       loss_sums[1] = sum over positives of smooth-L1(beta 1/9) / max(npos, 1); gout [N,4] = their
 *                       gradients with respect to out.  Keeps wgt [N,9], h [N,64] for the backward pass.
 *   sassd_aux_head_bwd  grad_sums [2] (device) = upstream gradients of the two sums -> grad_feats[s] [M_s, C_s]
 *                       (zeroed here, accumulated with float atomics like the reference's three_interpolate_grad),
 *                       dw1 [64,160], dw2 [4,64] (per-workgroup partials, fixed-order reduction).
 * Workspace (both): sassd_aux_head_workspace_bytes(N). */
size_t sassd_aux_head_workspace_bytes(int N);
int sassd_aux_prepare(const float *voxel_feats, int vstride, const int32_t *coors, int N, const int32_t *const *indices,
                      const int *M, const float *voxel_size, const float *offset, const float *gt_boxes,
                      const int32_t *gt_off, int B, float *points, float *const *known, uint8_t *label, float *target,
                      int *npos, void *stream);
int sassd_aux_head_fwd(int N, const float *const *feats, const int32_t *const *nn_idx, const float *const *nn_d2,
                       const float *w1, const float *w2, const uint8_t *label, const float *target, const int *npos,
                       float *wgt, float *h, float *out, float *gout, float *loss_sums, void *workspace,
                       size_t workspace_bytes, void *stream);
int sassd_aux_head_bwd(int N, const float *const *feats, const int *M, const int32_t *const *nn_idx, const float *w1,
                       const float *w2, const float *wgt, const float *h, const float *gout, const float *grad_sums,
                       float *const *grad_feats, float *dw1, float *dw2, void *workspace, size_t workspace_bytes,
                       void *stream);

/* dst[i] = map[i] >= 0 ? src[map[i]] : 0 for i < n; dst fp32, or bf16 (round-to-nearest-even) when bf16 != 0: every
 * kernel-layout weight image is a permutation (+ zero padding) of the flat parameter buffer, so ONE gather re-packs all
 * of them after an optimizer step (sassd.train.PackPlan). */
int sassd_gather_pack(const float *src, const int32_t *map, void *dst, long n, int bf16, void *stream);
int sassd_grad_sumsq(const float *grad, long n, float *out, void *stream);
int sassd_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n,
                    const float *grad_sumsq, float lr, float beta1, float beta2, float eps, float weight_decay,
                    int step, float max_norm, float grad_scale, void *stream);

/* Hardware self-test helper used by tests: D = A(32x2k) * B(2k x32) through v_mfma_f32_32x32x2_f32 and
 * D = A(16x4k) * B(4k x16) through v_mfma_f32_16x16x4_f32 with the lane maps the kernels assume. */
int sassd_mfma_probe(const float *a32, const float *b32, float *d32, const float *a16, const float *b16,
                     float *d16, int ksteps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SASSD_H */
