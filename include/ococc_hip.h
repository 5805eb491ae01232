/*
 * ococc_hip.h -- C ABI of libococc_hip.so, the MI355X (gfx950) kernels behind
 * the OcOccNet hot path of Ghostish/ObjectCentricOccCompletion.
 *
 * Drop-in boundary.  The reference binds this path through pybind11 torch
 * extensions that take at::Tensor (mmdet3d/ops/voxel/src/voxelization.cpp:6-11,
 * mmdet3d/ops/spconv/src/all.cc:21-51) plus three un-vendored CUDA packages
 * (TorchEx, torch_scatter, SpConv2).  This header replaces those bindings with
 * plain pointers and sizes; INTEGRATION.md shows the ctypes stub a reference
 * maintainer would add in the mmdet3d/ops python files to call it.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer unless the parameter is called host_*;
 *   - the caller owns every buffer, including `workspace` whose size comes from
 *     the matching *_workspace_bytes() query (pure host arithmetic);
 *   - nothing allocates, frees, synchronises or throws; work is queued on
 *     `stream` (a hipStream_t passed as void*, NULL = the null stream);
 *   - variable-size results are written into worst-case sized buffers and
 *     their length into a device int32 the caller reads back when it needs it;
 *   - return 0 on success, a negative OCOCC_E* code otherwise;
 *     ococc_last_error() returns the message of the calling thread's last
 *     failure;
 *   - row-major contiguous tensors; "bf16" buffers are uint16_t bit patterns.
 */
#ifndef OCOCC_HIP_H_
#define OCOCC_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCOCC_OK 0
#define OCOCC_EINVAL (-1)      /* bad argument (message says which) */
#define OCOCC_EHIP (-2)        /* HIP runtime error */
#define OCOCC_EUNSUPPORTED (-3) /* shape / dtype outside the compiled kernels */
#define OCOCC_ESTRANDED (-4)   /* a grid barrier of an earlier one-launch SIR layer gave up: its results are incomplete */

/* element types of feature buffers */
#define OCOCC_F32 0
#define OCOCC_BF16 1

/* reduce types; values follow reduce_t of the reference
 * (mmdet3d/ops/voxel/src/scatter_points_cuda.cu:6) */
#define OCOCC_REDUCE_SUM 0
#define OCOCC_REDUCE_MEAN 1
#define OCOCC_REDUCE_MAX 2

typedef void* ococc_stream_t; /* hipStream_t */

const char* ococc_last_error(void);
int ococc_version(void);
/* compiled for: returns "gfx950" */
const char* ococc_arch(void);

/* ------------------------------------------------------------------------ *
 * B1  dynamic voxelisation
 * replaces voxelization::dynamic_voxelize
 *   (mmdet3d/ops/voxel/src/voxelization.h:77-88, kernels
 *    voxelization_cuda.cu:25-65 / voxelization_cpu.cpp:8-41)
 * coors[i] = (z,y,x) int32, c = floor((p - min) / voxel) CLAMPED to
 * [0, grid-1] (this fork clamps; upstream mmdet3d writes -1),
 * grid = ceil((max - min) / voxel)  (voxelization_cpu.cpp:155-158).
 * points: [num_points, num_features] f32, first three columns are x,y,z.
 * ------------------------------------------------------------------------ */
int ococc_dynamic_voxelize_f32(const float* points, int64_t num_points, int32_t num_features,
                               const float host_voxel_size[3], const float host_coors_range[6],
                               int32_t* coors, ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * hard voxelisation (max_points / max_voxels caps, first-come order)
 * replaces voxelization::hard_voxelize
 *   (voxelization.h:60-75, voxelization_cpu.cpp:44-143).
 * grid = round((max - min) / voxel) (voxelization_cpu.cpp:119-122).
 * voxels [max_voxels,max_points,num_features] f32 (caller zero-fills),
 * coors [max_voxels,3] i32, num_points_per_voxel [max_voxels] i32 (zeroed),
 * voxel_num: device int32 receiving the number of voxels produced.
 * ------------------------------------------------------------------------ */
int64_t ococc_hard_voxelize_workspace_bytes(int64_t num_points, const float host_voxel_size[3],
                                            const float host_coors_range[6]);
int ococc_hard_voxelize_f32(const float* points, int64_t num_points, int32_t num_features,
                            const float host_voxel_size[3], const float host_coors_range[6],
                            int32_t max_points, int32_t max_voxels, float* voxels, int32_t* coors,
                            int32_t* num_points_per_voxel, int32_t* voxel_num, void* workspace,
                            int64_t workspace_bytes, ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * unique rows of integer coordinates on a bounded grid
 * replaces at::unique_dim(coors, 0, sorted, inverse, counts) as used by
 *   dynamic_point_to_voxel_forward (scatter_points_cuda.cu:199-210) and
 *   torch.unique(coors, dim=0, return_inverse=True) in scatter_v2
 *   (mmdet3d/ops/sst/sst_ops.py:150-181).
 * coors [n, ndim] int32 (ndim 1..4), host_dims[ndim] = exclusive upper bound
 * of each column.  Rows with any negative entry are dropped (inv = -1), as
 * DynamicScatter does; rows with an entry >= its bound set *status to 1.
 * Outputs are in lexicographic order of the rows (== torch.unique order):
 * out_coors [>= num_unique, ndim], inv [n] (row -> unique row), counts
 * [>= num_unique], num_unique (device int32).  out_capacity bounds the rows
 * written to out_coors/counts (use min(n, prod(dims))).
 * ------------------------------------------------------------------------ */
int64_t ococc_grid_unique_workspace_bytes(int32_t ndim, const int32_t host_dims[4]);
/* Byte offsets of the cell bitmap (1 bit per cell, row-major over dims) and of its exclusive popcount
 * prefix (one u32 per 32 cells) inside the grid_unique workspace: after ococc_grid_unique_i32 they
 * describe the sorted voxel set and can be handed to ococc_subm_rulebook_build_sorted. */
int ococc_grid_unique_workspace_layout(int32_t ndim, const int32_t host_dims[4], int64_t* bitmap_offset,
                                       int64_t* prefix_offset);
int ococc_grid_unique_i32(const int32_t* coors, int64_t n, int32_t ndim, const int32_t host_dims[4],
                          int32_t* out_coors, int64_t out_capacity, int32_t* inv, int32_t* counts,
                          int32_t* num_unique, int32_t* status, void* workspace,
                          int64_t workspace_bytes, ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * A5 / B2  segment (scatter) reduce, forward and backward
 * replaces feats_reduce_kernel + traceback kernels
 *   (scatter_points_cuda.cu:81-179) and torch_scatter.scatter_max /
 *   scatter(reduce='mean'|'sum') called at sst_ops.py:171-174.
 * feats [n, c] f32, inv [n] int32 in [-1, num_segments) (-1 rows ignored),
 * out [num_segments, c] f32.  counts [num_segments] int32 is required for
 * MEAN (divide) and optional otherwise (segments with count 0 -> 0 for MAX,
 * as torch_scatter does).  arg [num_segments, c] int32 (MAX only, may be
 * NULL): smallest row index attaining the max -- the tie rule of
 * max_reduce_traceback_scatter_idx_kernel (scatter_points_cuda.cu:136-160).
 * ------------------------------------------------------------------------ */
int ococc_segment_count_i32(const int32_t* inv, int64_t n, int32_t* counts, int64_t num_segments,
                            ococc_stream_t stream);
int ococc_segment_reduce_f32(const float* feats, const int32_t* inv, int64_t n, int32_t c,
                             int32_t reduce_type, const int32_t* counts, float* out, int32_t* arg,
                             int64_t num_segments, ococc_stream_t stream);
/* grad_feats [n, c] is fully written (no pre-zeroing needed).
 * SUM:  g[i] = go[inv[i]];  MEAN: g[i] = go[inv[i]] / counts[inv[i]];
 * MAX:  g[i][ch] = arg[inv[i]][ch] == i ? go[inv[i]][ch] : 0. */
int ococc_segment_reduce_bwd_f32(const float* grad_out, const int32_t* inv, int64_t n, int32_t c,
                                 int32_t reduce_type, const int32_t* counts, const int32_t* arg,
                                 float* grad_feats, int64_t num_segments, ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * B1 + B2 fused: dynamic voxelisation followed by DynamicScatter(mean)
 * replaces, for the per-object grid front end, the sequence
 *   voxelization::dynamic_voxelize      (mmdet3d/ops/voxel/src/voxelization.h:77-88)
 *   coors = cat(batch_idx, zyx)         (mmdet3d/models/detectors: every voxel pipeline)
 *   DynamicScatter.forward_single(mean) (mmdet3d/ops/voxel/scatter_points.py:53-107,
 *                                        scatter_points_cuda.cu:199-241)
 * with the results of running ococc_dynamic_voxelize_f32 + ococc_grid_unique_i32 +
 * ococc_segment_reduce_f32(MEAN) one after the other: voxel rows in ascending (b,z,y,x) order,
 * inv[i] = row of point i (-1: batch_idx[i] < 0, dropped), counts, and the mean of the point
 * features per voxel (sums of >= 3 points may differ in the last bit: float atomics, as there).
 * points [n, num_point_features] f32 (x,y,z first), batch_idx [n] int32 in [0, batch_size)
 * (>= batch_size sets *status = 1 and drops the point), feats [n, c] f32.
 * host_grid_zyx must equal ceil((max - min) / voxel) per axis (checked).
 * All out_capacity rows are written: rows past *num_voxels get -1 coordinates, count 0 and zero
 * features (the fixed-capacity form HIP-graph capture needs; use out_capacity = min(n, cells)).
 * voxel_feats [out_capacity, c] f32 is required (it is where the sums live); voxel_feats_bf16
 * (optional) receives the same means rounded to bf16 for the convolutions.
 * num_voxels / status: ONE device int32[2].  The workspace keeps the grid's cell bitmap and
 * popcount prefix at the offsets ococc_grid_unique_workspace_layout reports for
 * dims = {batch, D, H, W}, ready for ococc_subm_rulebook_build_sorted.
 * ------------------------------------------------------------------------ */
int64_t ococc_voxelize_scatter_workspace_bytes(int64_t n, int32_t batch_size, const int32_t host_grid_zyx[3]);
int ococc_voxelize_scatter_mean_f32(const float* points, int32_t num_point_features, const int32_t* batch_idx,
                                    int64_t n, const float* feats, int32_t c, const float host_voxel_size[3],
                                    const float host_coors_range[6], int32_t batch_size,
                                    const int32_t host_grid_zyx[3], int32_t* voxel_coors, int64_t out_capacity,
                                    int32_t* inv, int32_t* counts, float* voxel_feats, uint16_t* voxel_feats_bf16,
                                    int32_t* num_voxels, int32_t* status, void* workspace, int64_t workspace_bytes,
                                    ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * B1 + B2 + B3 for a batch of per-object grids in three launches: the outputs of
 * ococc_voxelize_scatter_mean_f32 (replaces Voxelization dynamic + DynamicScatter mean,
 * ops/voxel/voxelize.py:10-113, ops/voxel/scatter_points.py:53-107) AND of the 3x3x3 dilation-1
 * ococc_subm_rulebook_build_sorted on those voxels (replaces getIndicePair<3> with subM,
 * ops/spconv/include/spconv/spconv_ops.h:27-104 / geometry.h:247-297), bit for bit, in the
 * fixed-capacity form (every one of `capacity` rows is written; rows past *num_voxels carry -1
 * coordinates, count 0, zero features, -1 neighbours; indice_pairs[k][.][indice_num[k]..] is left
 * unwritten, as with fill_pair_tails = 0).
 * Preconditions beyond those of the two entry points: batch_idx is non-decreasing (points arrive
 * grouped by object grid; anything else sets *status = 1 and drops the offending points), the cells of
 * one grid are a multiple of 32 and at most 512 Ki (bitmap + prefix of one grid live in LDS),
 * capacity = min(n, batch * cells).  slices (1..8): workgroups per grid in the emit launch.
 * nbr_t [27, capacity] int32, blockmask [ceil(capacity/16)] u32, indice_pairs [27, 2, capacity] int32,
 * indice_num [27] int32.  The workspace starts with the grid_unique layout (bitmap | prefix), as the
 * voxelize_scatter workspace does.
 * ------------------------------------------------------------------------ */
int64_t ococc_object_grid_geometry_workspace_bytes(int64_t n, int32_t batch_size, const int32_t host_grid_zyx[3],
                                                   int32_t slices);
int ococc_object_grid_geometry_f32(const float* points, int32_t num_point_features, const int32_t* batch_idx,
                                   int64_t n, const float* feats, int32_t c, const float host_voxel_size[3],
                                   const float host_coors_range[6], int32_t batch_size,
                                   const int32_t host_grid_zyx[3], int32_t slices, int32_t* voxel_coors,
                                   int64_t capacity, int32_t* inv, int32_t* counts, float* voxel_feats,
                                   uint16_t* voxel_feats_bf16, int32_t* num_voxels, int32_t* status, int32_t* nbr_t,
                                   uint32_t* blockmask, int32_t* indice_pairs, int32_t* indice_num, void* workspace,
                                   int64_t workspace_bytes, ococc_stream_t stream);
/* The same, also building the neighbour-pattern row order (ococc_subm_row_order below) of the table it writes, where
 * the rows' 27 table entries sit in registers anyway: order_counters as there, order_rowrec [capacity, 4] int32, 16-byte
 * aligned (scratch: the per-row records), capacity below 2^20.
 *   order_rec [capacity, 4] int32 + order_hdr [8] int32 given: the finished order, as ococc_subm_row_order leaves it
 *     (heavy_blocks / mid_blocks as there): ococc_subm_row_order_place runs behind the last kernel.
 *   order_rec / order_hdr NULL: only the row records; follow with ococc_subm_row_order_place(order_rowrec, 27, 13,
 *     capacity, ...).
 * order_counters and order_rowrec NULL: exactly the call above. */
int ococc_object_grid_geometry_order_f32(const float* points, int32_t num_point_features, const int32_t* batch_idx,
                                         int64_t n, const float* feats, int32_t c, const float host_voxel_size[3],
                                         const float host_coors_range[6], int32_t batch_size,
                                         const int32_t host_grid_zyx[3], int32_t slices, int32_t* voxel_coors,
                                         int64_t capacity, int32_t* inv, int32_t* counts, float* voxel_feats,
                                         uint16_t* voxel_feats_bf16, int32_t* num_voxels, int32_t* status, int32_t* nbr_t,
                                         uint32_t* blockmask, int32_t* indice_pairs, int32_t* indice_num, void* workspace,
                                         int64_t workspace_bytes, void* order_counters, int32_t* order_rowrec,
                                         int32_t* order_rec, int32_t* order_hdr, int32_t heavy_blocks, int32_t mid_blocks,
                                         ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * visibility ray test of the GT-occupancy annotation
 * replaces point_cloud_to_range_image_idx (tools/occ/occ_annotate.py:141-207) and the gather /
 * compare / max-over-frames-and-sensors around it in OccAnnotator.annotate_trk (:488-556).
 * centers [n,3] f64: unoccupied cell centres in the object frame.  Frame f, sensor c (index
 * sf = c * frames + f):  p_ego = to_ego[f] (p),  p_sensor = to_sensor[sf] (p_ego)  with affines
 * stored as 12 doubles (row-major 3x3 rotation, then the translation; to_sensor = inverse of the
 * LiDAR extrinsic), az_corr[sf] = atan2(extrinsic[1][0], extrinsic[0][0]), inclinations
 * [sensors*frames, height] f64 in the order the range-image rows use (the reference flips the beam
 * table first), range_images[sf] -> device [height, width] f32 (range_dtype 0) or f64 (2).
 * visibility [n] int32 (optional): 2 where some sensor in some frame measured a range >= the
 * centre's range in the centre's pixel (the ray crossed the cell: empty), else 0.
 * dbg_indices [sensors*frames, n, 2] int32 (row, col) / dbg_range [sensors*frames, n] f64 (optional,
 * together): the outputs of point_cloud_to_range_image_idx, for parity tests.
 * ------------------------------------------------------------------------ */
int ococc_occ_visibility_f64(const double* centers, int64_t n, const double* to_ego, int32_t frames,
                             const double* to_sensor, const double* az_corr, const double* inclinations,
                             int32_t sensors, int32_t height, int32_t width, const void* const* range_images,
                             int32_t range_dtype, int32_t* visibility, int32_t* dbg_indices, double* dbg_range,
                             ococc_stream_t stream);

/* Same rulebook when the rows of `indices` are exactly the occupied cells of a [batch, D, H, W] grid in
 * ascending cell order and the caller still holds that grid's bitmap + popcount prefix (the state
 * ococc_grid_unique_i32 leaves in its workspace, see ococc_grid_unique_workspace_layout): the
 * marking, scanning and row-permutation passes are skipped.  Rows with negative coordinates
 * (fixed-capacity padding) take part in no pair.  fill_pair_tails = 0 leaves indice_pairs[k][.][indice_num[k]..]
 * unwritten (the reference format fills them with -1: 27 MB of stores per call that a fixed-capacity training
 * step never reads).  Workspace size as for ococc_subm_rulebook_build. */
int ococc_subm_rulebook_build_sorted(const int32_t* indices, int64_t n, int32_t batch_size,
                                     const int32_t host_shape[3], const int32_t host_ksize[3],
                                     const uint32_t* grid_bitmap, const uint32_t* grid_prefix, int32_t* nbr_t,
                                     uint32_t* blockmask, int32_t* indice_pairs, int32_t* indice_num,
                                     int32_t fill_pair_tails, void* workspace, int64_t workspace_bytes,
                                     ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * B3  sub-manifold rulebook
 * replaces spconv::getIndicePair<3>(subM) (include/spconv/spconv_ops.h:27-104
 * -> geometry.h:247-297 CPU / indice.cu.h:147-234 GPU).
 * indices [n, 4] int32 (batch, z, y, x); spatial shape host_shape[3] (D,H,W);
 * kernel size host_ksize[3] (odd), dilation host_dilation[3].
 * Products:
 *   nbr_t     [kvol, n] int32  nbr_t[k][o] = input row feeding output row o
 *                              through kernel offset k, or -1;
 *   blockmask [ceil(n/16)] u32 bit k set iff rows 16b..16b+15 have any
 *                              neighbour at offset k (kvol <= 32 only,
 *                              otherwise pass NULL);
 *   indice_pairs [kvol,2,n] int32 and indice_num [kvol] int32 -- the
 *     reference rulebook, filled with -1 beyond indice_num[k], in the exact
 *     order of the CPU functor (ascending input row within an offset).
 * offset index k = sum_d m_d * (in_d - out_d + pad_d), last dim fastest
 * (geometry.h:61-72).
 * ------------------------------------------------------------------------ */
int64_t ococc_subm_rulebook_workspace_bytes(int64_t n, int32_t batch_size,
                                            const int32_t host_shape[3],
                                            const int32_t host_ksize[3]);
int ococc_subm_rulebook_build(const int32_t* indices, int64_t n, int32_t batch_size,
                              const int32_t host_shape[3], const int32_t host_ksize[3],
                              const int32_t host_dilation[3], int32_t* nbr_t, uint32_t* blockmask,
                              int32_t* indice_pairs, int32_t* indice_num, void* workspace,
                              int64_t workspace_bytes, ococc_stream_t stream);

/* Regular (strided) / transposed sparse conv rulebook: the non sub-manifold branch of
 * spconv::getIndicePair (spconv_ops.h:105-141; geometry.h:144-245, indice.cu.h:22-145).
 * Active outputs are numbered in sorted order of their flat grid index (the reference's GPU
 * path; its CPU functor numbers by first appearance); pairs of an offset are in ascending
 * input row.  out_indices [out_capacity,4] (use n*kvol as the safe bound or
 * min(n*kvol, batch*D*H*W)), indice_pairs [kvol,2,n] (-1 filled), indice_num [kvol],
 * num_out: device int32.  kvol <= 255. */
int64_t ococc_conv_rulebook_workspace_bytes(int64_t n, int32_t batch_size,
                                            const int32_t host_out_shape[3],
                                            const int32_t host_ksize[3]);
int ococc_conv_rulebook_build(const int32_t* indices, int64_t n, int32_t batch_size,
                              const int32_t host_out_shape[3], const int32_t host_ksize[3],
                              const int32_t host_stride[3], const int32_t host_padding[3],
                              const int32_t host_dilation[3], int32_t transpose,
                              int32_t* out_indices, int64_t out_capacity, int32_t* indice_pairs,
                              int32_t* indice_num, int32_t* num_out, void* workspace,
                              int64_t workspace_bytes, ococc_stream_t stream);

/* Rebuild the gather tables from a reference-format rulebook (any conv type:
 * regular / inverse / user supplied).  Each (row, offset) may appear at most
 * once on the chosen side (true for every spconv rulebook).
 * side = 1: table[k][pairs[k][1][p]] = pairs[k][0][p]  (forward: out <- in)
 * side = 0: table[k][pairs[k][0][p]] = pairs[k][1][p]  (dgrad:   in  <- out)
 * table [kvol, num_rows] int32 is fully written (-1 where empty). */
int ococc_rulebook_pairs_to_table(const int32_t* indice_pairs, const int32_t* indice_num,
                                  int32_t kvol, int64_t pair_capacity, int32_t side,
                                  int64_t num_rows, int32_t* table, uint32_t* blockmask,
                                  ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * B4  sparse convolution:  out[o] = sum_k  feat[table[k][o]] @ W[k]
 * replaces indiceConv / indiceConvBackward (spconv_ops.h:260-456): the
 * reference runs, per offset, gather kernel + GEMM + scatter-add kernel
 * (reordering.cu.h:21-157); here one output-stationary kernel gathers rows
 * straight into MFMA operand registers with the weights resident in LDS.
 *
 * feat   [n_in, kd] bf16        kd multiple of 16, <= 256
 * wn     [kvol, ncols, kd] bf16 weights with the contraction dim innermost
 *                               (forward: W[k]^T; dgrad: W[k'] as stored, see
 *                               ococc_weight_prepare_bf16)
 * table  [kvol, n_out] int32, blockmask [ceil(n_out/16)] (kvol <= 32)
 * bias   [ncols] f32 or NULL
 * out    [n_out, ncols] bf16 (out_dtype OCOCC_BF16) or f32; ncols mult. of 16
 * ------------------------------------------------------------------------ */
int ococc_sparse_conv_gather_gemm_bf16(const uint16_t* feat, int64_t n_in, int32_t kd,
                                       const uint16_t* wn, int32_t kvol, int32_t ncols,
                                       const int32_t* table, const uint32_t* blockmask,
                                       int64_t n_out, const float* bias, void* out,
                                       int32_t out_dtype, ococc_stream_t stream);

/* The same convolution for SUB-MANIFOLD tables over sparse active sets (a voxel has only a few
 * neighbours): the centre offset dense_k (every row is its own neighbour; -1: none) is a dense pass,
 * every other offset's rows are compacted to the ones that do have a neighbour before they are
 * gathered and multiplied, with f32 accumulators of a 512-row tile in LDS.  Same operands and the
 * same result (f32 accumulation, centre first, then ascending offsets: deterministic) as
 * ococc_sparse_conv_gather_gemm_bf16; faster below ~2-3 rulebook pairs per output row
 * (tools/density_sweep.py), slower on dense neighbourhoods.  kd in {32, 64, 128}, ncols in {32, 64, 128}; no block masks needed. */
int ococc_sparse_conv_tile_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn, int32_t kvol,
                                int32_t ncols, const int32_t* table, int32_t dense_k, int64_t n_out,
                                const float* bias, void* out, int32_t out_dtype, ococc_stream_t stream);
/* The same convolution with the OUTPUT rows processed in neighbour-pattern order (sub-manifold tables, kvol <= 32).
 * ococc_subm_row_order buckets the rows of an offset-major gather table by (number of neighbours besides dense_k: 3+,
 * 2, 1, 0; lowest two neighbour offsets) -- rows that share their offsets become neighbours in the order, so a
 * workgroup walks only the offsets its rows have and 16-row MFMA blocks are nearly full.  The order is a property of
 * the table: build it once per rulebook, use it for every layer and direction that gathers through the table.
 *   counters ococc_subm_row_order_counter_bytes() bytes of scratch: cleared by every build before it counts (a memset node
 *            here; inside the first geometry kernel in ococc_object_grid_geometry_order_f32), contents undefined after
 *   scratch  ococc_subm_row_order_scratch_bytes(n) bytes, 16-byte aligned (the row records: n below 2^20)
 *   rec      [n, 4] int32, 16-byte aligned: per slot {row, offset mask (bit k: table[k][row] >= 0), table entries at the
 *            row's lowest and second lowest neighbour offsets (-1: none)}
 *   hdr      [8] int32   tile plan read by the kernel (heavy_blocks / mid_blocks: 16-row blocks per workgroup tile
 *                        for the 3+ / 2 neighbour classes: 4, 8 or 16; the rest uses 16)
 * Slots inside a bucket are handed out with atomics: the order may differ between two builds, the convolution's
 * result does not (each output row accumulates its own products in ascending offset order, in f32).
 * ococc_sparse_conv_sorted_bf16: operands as ococc_sparse_conv_gather_gemm_bf16 (wn row-major [kvol, ncols, kd]), same
 * result bit for bit; kd, ncols in {32, 64, 128}, not both 128.  Replaces indiceConv's per-offset gather / GEMM /
 * scatter-add (spconv_ops.h:300-354). */
int64_t ococc_subm_row_order_counter_bytes(void);
int64_t ococc_subm_row_order_scratch_bytes(int64_t n);
int ococc_subm_row_order(const int32_t* table, int32_t kvol, int32_t dense_k, int64_t n, int32_t heavy_blocks,
                         int32_t mid_blocks, void* counters, void* scratch, int32_t* rec, int32_t* hdr,
                         ococc_stream_t stream);
/* The second half of ococc_subm_row_order alone, for row records some other kernel left while it wrote the table
 * (ococc_object_grid_geometry_order_f32 without order_rec, which also counted them into ``counters``): records ->
 * slots, header. */
int ococc_subm_row_order_place(const int32_t* rowrec, int32_t kvol, int32_t dense_k, int64_t n, int32_t heavy_blocks,
                               int32_t mid_blocks, void* counters, int32_t* rec, int32_t* hdr, ococc_stream_t stream);
int ococc_sparse_conv_sorted_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn, int32_t kvol,
                                  int32_t ncols, const int32_t* table, const int32_t* rec, const int32_t* hdr,
                                  int64_t n_out, const float* bias, void* out, int32_t out_dtype, ococc_stream_t stream);
/* ococc_sparse_conv_sorted_bf16 with the LayerNorm (+ GELU) of the enclosing conv -> norm -> act block
 * (make_sparse_convmodule, mmdet3d/ops/sparse_block.py:216-289) in its epilogue: the contract of
 * ococc_sparse_conv_tile_ln_bf16 below (conv_out, y = act(LN(conv_out)), mean_rstd [n_out, 2]) for rulebooks that run in
 * neighbour-pattern order.  The finished f32 row sits in the registers of the four lanes that share a slot; the norm
 * sees the bf16-rounded conv output and sums in the order of ococc_layernorm_act_fwd, so y and mean_rstd equal the
 * two-launch pair's (conv, then LN).  kd, ncols in {32, 64, 128}, not both 128 (round 6: the 64 -> 128 forward of
 * configs[1], whose separate LN launch re-read 32 MB). */
int ococc_sparse_conv_sorted_ln_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn, int32_t kvol,
                                     int32_t ncols, const int32_t* table, const int32_t* rec, const int32_t* hdr,
                                     int64_t n_out, const float* gamma, const float* beta, float eps, int32_t act,
                                     uint16_t* conv_out, uint16_t* y, float* mean_rstd, ococc_stream_t stream);
/* ococc_sparse_conv_sorted_bf16 as the input-gradient pass of layer L+1 with the LayerNorm (+ GELU) BACKWARD of the
 * conv -> LN -> act block L in its epilogue -- the contract of ococc_sparse_conv_tile_lnbwd_bf16 below (same operands
 * besides the row order rec / hdr and the row-major dgrad operand wn; d_conv_out bit-identical to it and to
 * ococc_layernorm_act_bwd on the bf16 dgrad output), for rulebooks that run in neighbour-pattern order.  One row of
 * partial sums [d gamma | d beta] per workgroup: partial_rows >= ococc_sparse_conv_sorted_lnbwd_partial_rows(n_out),
 * every row written.  ncols in {32, 64}.  Replaces indiceConvBackward's input-gradient loop (spconv_ops.h:363-456)
 * followed by the LayerNorm backward of sparse_block.py:216-289's norm layer. */
int64_t ococc_sparse_conv_sorted_lnbwd_partial_rows(int64_t n_out);
int ococc_sparse_conv_sorted_lnbwd_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn, int32_t kvol,
                                        int32_t ncols, const int32_t* table, const int32_t* rec, const int32_t* hdr,
                                        int64_t n_out, const uint16_t* block_conv_out, const float* mean_rstd,
                                        const float* gamma, const float* beta, int32_t act, uint16_t* d_conv_out,
                                        float* partials, int64_t partial_rows, ococc_stream_t stream);
/* The same kernel with the LayerNorm (+ GELU) that follows the convolution in the reference's
 * make_sparse_convmodule block (mmdet3d/ops/sparse_block.py:216-289: conv -> LN(eps) -> GELU) applied in the
 * epilogue, where the finished f32 row sits in LDS: conv_out [n_out, ncols] bf16 (what the LN backward needs), y =
 * act(LN(conv_out)) bf16, mean_rstd [n_out, 2] f32.  The norm sees the bf16-rounded conv output, as the separate
 * ococc_layernorm_act_fwd would.  Shapes up to 64 x 64 channels; OCOCC_EUNSUPPORTED otherwise. */
int ococc_sparse_conv_tile_ln_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn, int32_t kvol,
                                   int32_t ncols, const int32_t* table, int32_t dense_k, int64_t n_out,
                                   const float* gamma, const float* beta, float eps, int32_t act,
                                   uint16_t* conv_out, uint16_t* y, float* mean_rstd, ococc_stream_t stream);
/* The same kernel as the input-gradient pass of layer L+1 (feat = d conv_out of layer L+1, wn = its dgrad operand,
 * table = its gather-side table for the backward direction) with the LayerNorm (+ GELU) BACKWARD of the block in
 * front (layer L: conv -> LN -> act, sparse_block.py:216-289) applied in the epilogue: the finished row is the
 * gradient of layer L's block OUTPUT; with layer L's saved conv output block_conv_out [n_out, ncols] bf16, its
 * statistics mean_rstd [n_out, 2] and gamma / beta, d_conv_out [n_out, ncols] bf16 = the gradient of layer L's CONV
 * output leaves instead (bit-identical to ococc_layernorm_act_bwd on the bf16 dgrad output), plus one row of
 * partial sums [d gamma | d beta] per workgroup into partials [partial_rows, 2 ncols] f32 (every row written;
 * ococc_layernorm_param_reduce_multi or a column sum finishes them).  ncols in {32, 64}. */
int64_t ococc_sparse_conv_tile_lnbwd_partial_rows(int64_t n_out, int32_t kd, int32_t ncols);
int ococc_sparse_conv_tile_lnbwd_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn, int32_t kvol,
                                      int32_t ncols, const int32_t* table, int32_t dense_k, int64_t n_out,
                                      const uint16_t* block_conv_out, const float* mean_rstd, const float* gamma,
                                      const float* beta, int32_t act, uint16_t* d_conv_out, float* partials,
                                      int64_t partial_rows, ococc_stream_t stream);

/* weights [kvol, cin, cout] (the reference layout (kD,kH,kW,Cin,Cout),
 * spconv/conv.py:98-99) in f32 or bf16 ->
 *   mode 0 (forward): wn[k][cout][cin]  = W[k][cin][cout]
 *   mode 1 (dgrad, sub-manifold): wn[k][cin][cout] = W[kvol-1-k][cin][cout]
 *   mode 2 (dgrad, generic table side 0): wn[k][cin][cout] = W[k][cin][cout]
 * always bf16 out. */
/* Same conversion for up to 16 f32 weight tensors in ONE launch (host arrays of device pointers and
 * sizes; argument meaning per entry as ococc_weight_prepare_bf16). */
int ococc_weight_prepare_multi_bf16(int32_t count, const void* const* w, const int32_t* kvol, const int32_t* cin,
                                    const int32_t* cout, const int32_t* mode, void* const* wn,
                                    ococc_stream_t stream);

/* Forward gather-GEMM with the norm/activation pair of make_sparse_convmodule
 * (mmdet3d/ops/sparse_block.py:216-289: conv -> LayerNorm -> GELU) fused into the epilogue:
 * conv_out [n_out, ncols] bf16 (kept for the backward), y = act(LN(conv_out)) bf16, mean_rstd [n_out,2]
 * f32 as ococc_layernorm_act_fwd writes them.  gamma / beta [ncols] f32.  Returns OCOCC_EUNSUPPORTED
 * when the shape has no fused kernel (then call the two separate entry points). */
int ococc_sparse_conv_gather_gemm_ln_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn,
                                          int32_t kvol, int32_t ncols, const int32_t* table,
                                          const uint32_t* blockmask, int64_t n_out, const float* gamma,
                                          const float* beta, float eps, int32_t act, uint16_t* conv_out,
                                          uint16_t* y, float* mean_rstd, ococc_stream_t stream);

int ococc_weight_prepare_bf16(const void* w, int32_t w_dtype, int32_t kvol, int32_t cin,
                              int32_t cout, int32_t mode, uint16_t* wn, ococc_stream_t stream);

/* dW[k] = sum_p x[pairs[k][0][p]]^T dy[pairs[k][1][p]]  over the rulebook.
 * x [n_in, cin] bf16, dy [n_out, cout] bf16, cin/cout multiples of 16,
 * dw [kvol, cin, cout] f32 (fully written).  Deterministic: partial sums go
 * to per-workgroup slabs in `workspace` and are added in a fixed order. */
int64_t ococc_sparse_conv_wgrad_workspace_bytes(int32_t kvol, int64_t pair_capacity, int32_t cin,
                                                int32_t cout);
int ococc_sparse_conv_wgrad_bf16(const uint16_t* x, int64_t n_in, int32_t cin, const uint16_t* dy,
                                 int64_t n_out, int32_t cout, const int32_t* indice_pairs,
                                 const int32_t* indice_num, int32_t kvol, int64_t pair_capacity,
                                 float* dw, void* workspace, int64_t workspace_bytes,
                                 ococc_stream_t stream);
/* Sparse max pooling over a rulebook -- replaces indice_maxpool_fp32/half and indice_maxpool_backward_fp32/half
 * (mmdet3d/ops/spconv/ops.py:162-184, src/maxpool.cc:9-55).  The reference's semantics: the output starts at ZERO and
 * takes an input only where that is larger (an output whose inputs are all negative holds 0); the backward pass gives
 * an output's gradient to every input equal to it.  table [kvol, n_out]: input row of (offset, output row) or -1
 * (ococc_rulebook_pairs_to_table, side of the forward direction); table_bwd [kvol, n_in]: output row of (offset, input
 * row) or -1.  dtype OCOCC_F32 / OCOCC_BF16 for features, out, out_bp and input_bp alike.  Sums in ascending offset
 * order, as the reference's CPU functor: f32 results bit for bit. */
int ococc_indice_maxpool(const void* features, int32_t dtype, int64_t n_in, int32_t channels, const int32_t* table,
                         int32_t kvol, int64_t n_out, void* out, ococc_stream_t stream);
int ococc_indice_maxpool_backward(const void* features, const void* out_features, const void* out_bp, int32_t dtype,
                                  int64_t n_in, int32_t channels, const int32_t* table_bwd, int32_t kvol, int64_t n_out,
                                  void* input_bp, ococc_stream_t stream);
/* The slab pass of ococc_sparse_conv_wgrad_bf16 (dw == NULL form) for TWO or THREE layers in one launch: arrays of
 * ``count``, shapes 16 x 32, 32 x 64 and 64 x 128 (OCOCC_EUNSUPPORTED otherwise; put the longest first).  Each is a few
 * hundred latency-bound work items that fill a fifth of the chip: side by side they take little more than the longest.
 * Same slabs as separate calls, bit for bit; finish them with ococc_sparse_conv_wgrad_reduce_multi /
 * ococc_backward_param_reduce_multi. */
int ococc_sparse_conv_wgrad_multi_bf16(int32_t count, const uint16_t* const* x, const uint16_t* const* dy, const int32_t* cin,
                                       const int32_t* cout, const int32_t* const* indice_pairs,
                                       const int32_t* const* indice_num, const int32_t* kvol, const int64_t* pair_capacity,
                                       void* const* workspaces, const int64_t* workspace_bytes, ococc_stream_t stream);
/* Both kinds of end-of-backward parameter-gradient sums in ONE launch: the weight-gradient slab sums of
 * ococc_sparse_conv_wgrad_reduce_multi (first five tables) and the LayerNorm d gamma / d beta column sums of
 * ococc_layernorm_param_reduce_multi (last five); same results as the two calls, bit for bit.  (No reference
 * counterpart: indiceConvBackward and torch's native_layer_norm_backward reduce inside their own kernels.) */
int ococc_backward_param_reduce_multi(int32_t wcount, const void* const* workspaces, const int32_t* const* indice_nums,
                                      const int32_t* kvols, const int64_t* elems, float* const* dws, int32_t lcount,
                                      const void* const* partials, const int32_t* rows, const int32_t* c,
                                      void* const* dgamma, void* const* dbeta, ococc_stream_t stream);
/* dw == NULL above stops after the per-work-item slabs (left in `workspace`); this call finishes up to 8 such
 * weight gradients in one launch: dw[i][k][e] = sum of the slabs of offset k, fixed order (deterministic).
 * elems[i] = cin * cout of layer i.  Used to move the reductions of a backward pass behind its last kernel. */
int ococc_sparse_conv_wgrad_reduce_multi(int32_t count, const void* const* workspaces,
                                         const int32_t* const* indice_nums, const int32_t* kvols,
                                         const int64_t* elems, float* const* dws, ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * B5  fused LayerNorm + GELU(erf) over feature rows (the norm/act pair that
 * make_sparse_convmodule appends to a sparse conv, ops/sparse_block.py:216-289;
 * also every Linear->LN->GELU of build_mlp, sst_ops.py:333-360).
 * x, y [n, c] in `dtype` (f32 or bf16); gamma, beta [c] f32.
 * act: 0 = none (plain LayerNorm), 1 = GELU(erf).
 * mean_rstd [n, 2] f32 is written by forward and read by backward.
 * Backward: dx [n,c] in dtype; dgamma, dbeta [c] f32 are OVERWRITTEN with the sum
 * of per-workgroup partials taken in a fixed order (deterministic).
 * ------------------------------------------------------------------------ */
int ococc_layernorm_act_fwd(const void* x, int64_t n, int32_t c, const float* gamma,
                            const float* beta, float eps, int32_t act, void* y, float* mean_rstd,
                            int32_t dtype, ococc_stream_t stream);
int64_t ococc_layernorm_act_bwd_workspace_bytes(int64_t n, int32_t c);
int ococc_layernorm_act_bwd(const void* x, const void* dy, int64_t n, int32_t c,
                            const float* gamma, const float* beta, const float* mean_rstd,
                            int32_t act, void* dx, float* dgamma, float* dbeta, int32_t dtype,
                            void* workspace, int64_t workspace_bytes, ococc_stream_t stream);
/* The same two kernels with the Dropout that follows the activation in build_mlp's Sequential(Linear, norm, act,
 * Dropout) (mmdet3d/ops/sst/sst_ops.py:333-360; occ_dropout = 0.1 in configs/ococc/ococcnet.py) folded in: y =
 * dropout(act(LN(x))).  The keep mask is a counter-based hash of (seed, element index), 16 bits per element against
 * drop_threshold = round(p * 65536) (0: no dropout; kept values are scaled by 65536 / (65536 - drop_threshold)); the
 * backward regenerates it from the same (drop_threshold, seed), no mask is stored.  bf16 rows of 8 * 2^k <= 512 or
 * 1024 / 1536 / 2048 channels; OCOCC_EUNSUPPORTED otherwise.  (torch's own Philox stream cannot be reproduced by a
 * different kernel anyway; the statistics are what is kept.) */
int ococc_layernorm_act_dropout_fwd_bf16(const uint16_t* x, int64_t n, int32_t c, const float* gamma, const float* beta,
                                         float eps, int32_t act, uint32_t drop_threshold, uint64_t seed, uint16_t* y,
                                         float* mean_rstd, ococc_stream_t stream);
int ococc_layernorm_act_dropout_bwd_bf16(const uint16_t* x, const uint16_t* dy, int64_t n, int32_t c,
                                         const float* gamma, const float* beta, const float* mean_rstd, int32_t act,
                                         uint32_t drop_threshold, uint64_t seed, uint16_t* dx, float* dgamma,
                                         float* dbeta, void* workspace, int64_t workspace_bytes, ococc_stream_t stream);
/* dgamma == dbeta == NULL in ococc_layernorm_act_bwd: dx and the per-block partial sums only (the workspace
 * then holds ococc_layernorm_act_bwd_partial_rows(n, c, dtype) rows of [2c] f32 and must stay alive).
 * ococc_layernorm_param_reduce_multi finishes up to 16 such layers in ONE launch -- the host mirror queues it
 * to the end of the autograd backward pass (torch's LayerNorm backward, which the reference's
 * build_norm_layer(dict(type='LN')) layers use, returns the parameter gradients per layer:
 * mmdet3d/ops/sparse_block.py:216-289). */
int32_t ococc_layernorm_act_bwd_partial_rows(int64_t n, int32_t c, int32_t dtype);
int ococc_layernorm_param_reduce_multi(int32_t count, const void* const* partials, const int32_t* rows,
                                       const int32_t* c, void* const* dgamma, void* const* dbeta,
                                       ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * A3  points-in-rotated-box pooling
 * replaces TorchEx dynamic_point_pool_ext.dynamic_point_pool_mixed_gpu as called at
 *   mmdet3d/ops/dynamic_point_pool_op.py:81-87 (source not vendored; contract from the
 *   call site and the debug assertions of
 *   models/roi_heads/roi_extractors/dynamic_point_roi_extractor.py:222-234).
 * rois [R,7] f32 (x,y,z_bottom,w,l,h,yaw), rois_key [R] i32, pts [N,3] f32, pts_key [N] i32.
 * A pair (point, RoI) is emitted when the keys match and the point is inside the RoI
 * enlarged by host_extra_wlh (w,l,h; extra/2 per side).  Rows are written SORTED by
 * (RoI, point index): out_pts_idx / out_roi_idx [>= num_out] i64, out_pts_feats
 * [>= num_out, 13] f32 = xyz, box-frame xyz, 6 face distances of the original box,
 * is_in_margin.  Caps: max_inbox_point per RoI, max_all_pts rows in total (the smallest
 * point indices / RoI indices are kept).  roi_counts [R] i32 (may be NULL) = rows kept per
 * RoI; num_out = rows written (device int32).
 * ------------------------------------------------------------------------ */
int64_t ococc_point_pool_workspace_bytes(int64_t num_points, int64_t num_rois);
int ococc_dynamic_point_pool_mixed(const float* rois, const int32_t* rois_key, int64_t num_rois,
                                   const float* pts, const int32_t* pts_key, int64_t num_points,
                                   const float host_extra_wlh[3], int32_t max_inbox_point,
                                   int64_t max_all_pts, int64_t* out_pts_idx, int64_t* out_roi_idx,
                                   float* out_pts_feats, int32_t* roi_counts, int32_t* num_out,
                                   void* workspace, int64_t workspace_bytes, ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * A2  pairwise (i-th with i-th) 3-D IoU of rotated boxes
 * replaces LiDARInstance3DBoxes.aligned_iou_3d
 *   (mmdet3d/core/bbox/structures/lidar_box3d.py:404-448 -> TorchEx boxes_overlap_1to1).
 * boxes [n,7] f32 (x, y, z_bottom, w, l, h, yaw); iou [n] f32.
 * ------------------------------------------------------------------------ */
int ococc_aligned_iou3d_f32(const float* boxes1, const float* boxes2, int64_t n, float* iou,
                            ococc_stream_t stream);

/* ------------------------------------------------------------------------
 * A12 glue, one launch each (f32; the element-wise chains they replace were 12-35 launches of a few hundred elements):
 * ococc_rotate_z_f32: rotation_3d_in_axis(points [n, m, 3], angles [n], axis=2)
 *   (mmdet3d/core/bbox/structures/utils.py:21-61; the reference's transposed-matrix convention: x' = x c + y s, y' = -x s + y c).
 * ococc_points_box_to_box_f32: points [n, m, 3] given in the frame of from_boxes[i] (gravity centred) -> the frame of
 *   to_boxes[i]: rotate by yaw_from, + centre_from, z + h_from / 2, - centre_to, z - h_to / 2, rotate by -yaw_to
 *   (ococc_bbox_head.py:1279-1290, 714-724; boxes are rows of ld >= 7 floats: x, y, z_bottom, w, l, h, yaw).
 * ococc_roi_box_targets_f32: GT boxes in the canonical frame of their RoIs -> DeltaXYZWLHRBBoxCoder deltas [n, 7]
 *   (ococc_bbox_head.py:1190-1222, delta_xyzwhlr_bbox_coder.py:21-50). */
int ococc_rotate_z_f32(const float* points, const float* angles, int64_t n, int64_t m, float* out, ococc_stream_t stream);
int ococc_points_box_to_box_f32(const float* points, const float* from_boxes, int64_t ld_from, const float* to_boxes,
                                int64_t ld_to, int64_t n, int64_t m, float* out, ococc_stream_t stream);
int ococc_roi_box_targets_f32(const float* rois, int64_t ld_rois, const float* gt_boxes, int64_t ld_gt, int64_t n, float* out,
                              ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * B6  in-group ranks of integer keys (SST window bookkeeping)
 * replaces TorchEx ingroup_indices.forward (mmdet3d/ops/sst/sst_ops.py:243-263; python twin
 * get_inner_win_inds_deprecated :194-241) and make_continuous_inds (:316-330).
 * keys [n] int32 in [0, key_bound) (negative keys are skipped: conti = inner = -1).
 * conti [n]: rank of the key among the distinct keys; inner [n]: number of earlier elements
 * with the same key (stable); counts [>= num_groups] (buffer of n ints, may be NULL);
 * num_groups, status (1 if a key >= key_bound was seen): device int32.
 * ------------------------------------------------------------------------ */
int64_t ococc_group_rank_workspace_bytes(int64_t n, int64_t key_bound);
int ococc_group_rank_i32(const int32_t* keys, int64_t n, int64_t key_bound, int32_t* conti, int32_t* inner,
                         int32_t* counts, int32_t* num_groups, int32_t* status, void* workspace,
                         int64_t workspace_bytes, ococc_stream_t stream);

/* ------------------------------------------------------------------------ *
 * B7  window attention core  out = softmax(q k^T * scale + key mask) v  per (window, head)
 * replaces the attention inside nn.MultiheadAttention as WindowAttention calls it on padded
 * [num_windows, max_tokens, C] tensors (mmdet3d/models/sst/sst_basic_block_v2.py:41-75).
 * q, k, v: bf16 rows (window*max_tokens + token) with the given row strides (elements), head h in
 * columns [16h, 16h+16); key_len [num_windows] int32 = valid tokens (a prefix) of each window;
 * out bf16 like q; lse [num_windows, num_heads, max_tokens] f32 (log-sum-exp of the scaled
 * scores, saved for backward).  head_dim must be 16, max_tokens <= 160.
 * ------------------------------------------------------------------------ */
int ococc_window_attn_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_stride,
                               int64_t k_stride, int64_t v_stride, const int32_t* key_len,
                               int64_t num_windows, int32_t max_tokens, int32_t num_heads, int32_t head_dim,
                               float scale, uint16_t* out, int64_t out_stride, float* lse,
                               ococc_stream_t stream);
int ococc_window_attn_bwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_stride,
                               int64_t k_stride, int64_t v_stride, const uint16_t* out, const uint16_t* dout,
                               int64_t o_stride, const float* lse, const int32_t* key_len,
                               int64_t num_windows, int32_t max_tokens, int32_t num_heads, int32_t head_dim,
                               float scale, uint16_t* dq, uint16_t* dk, uint16_t* dv, int64_t dq_stride,
                               int64_t dk_stride, int64_t dv_stride, ococc_stream_t stream);

/* Flat-token forms of the two calls above: q/k/v/out (and dout, dq/dk/dv) are [num_tokens, .] tensors in
 * the model's token order and token_index [num_windows * max_tokens] int32 names the row of every window
 * slot (-1 = padding; valid slots are a prefix of each window, key_len of them).  The kernel gathers a
 * window's tokens itself and writes every output row exactly once, so the padded [nW, T, C] copies that
 * flat2window / window2flat build in the reference (sst_ops.py:66-148) never exist. lse stays [nW, H, T]. */
int ococc_window_attn_fwd_gather_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_stride,
                                      int64_t k_stride, int64_t v_stride, const int32_t* token_index,
                                      const int32_t* key_len, int64_t num_windows, int32_t max_tokens,
                                      int32_t num_heads, int32_t head_dim, float scale, uint16_t* out,
                                      int64_t out_stride, float* lse, ococc_stream_t stream);
int ococc_window_attn_bwd_gather_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_stride,
                                      int64_t k_stride, int64_t v_stride, const uint16_t* out, const uint16_t* dout,
                                      int64_t o_stride, const float* lse, const int32_t* token_index,
                                      const int32_t* key_len, int64_t num_windows, int32_t max_tokens,
                                      int32_t num_heads, int32_t head_dim, float scale, uint16_t* dq, uint16_t* dk,
                                      uint16_t* dv, int64_t dq_stride, int64_t dk_stride, int64_t dv_stride,
                                      ococc_stream_t stream);

/* ------------------------------------------------------------------------
 * B7, fused  one SST encoder layer as two tile kernels per direction
 * replaces EncoderLayer.forward (post-norm, mmdet3d/models/sst/sst_basic_block_v2.py:105-127):
 *   window_attn_block:  y1 = norm1(x + self_attn(q = k = x + pos, v = x))   (WindowAttention.forward :41-75 around
 *                        nn.MultiheadAttention on flat2window_v2 / window2flat_v2 copies, sst_ops.py:66-148)
 *   token_ffn_block:    y2 = norm2(y1 + linear2(act(linear1(y1))))
 * d_model 128, 8 heads of 16, feed-forward 256; tokens bf16 [num_tokens, 128] in the model's flat order; all sums f32.
 *
 * Weights arrive as MFMA A-operand fragments: ococc_linear_fragments_bf16 turns f32 matrices S (rows x cols, any
 * element strides -- a transposed view is just swapped strides) into bf16 [rows/16][cols/32][64 lanes][8]:
 * lane 16 g + r of block (rb, cs) holds S[16 rb + r][32 cs + 8 g .. + 7].  For y = x W^T pass W ([out, in] as
 * nn.Linear stores it); for the input gradient dx = dy W pass W^T (rows = in, cols = out).
 *
 * Windows -> tiles: ococc_window_tile_plan packs whole windows (win_len[w] <= 64 tokens, their flat rows at
 * tok[win_off[w] .. + win_len[w])) greedily, in the given order, into tiles of 64 token slots: tile_rows [tiles*64]
 * = flat row of the slot or -1, tile_span [tiles*64] = lo | hi << 8, the slots of the slot's window (0: empty).
 * cap_tiles >= num_windows rows of both arrays are initialised; *num_tiles (device) receives the tiles in use.
 * A slot attends exactly the slots of its span: the key_padding_mask of the reference is the span.
 *
 * Backward kernels recompute their block from its input (nothing else is saved by the forward) and also write the
 * operands of the weight gradients -- attention block: dqkv [*,384] (gradients of q | k | v), attn_out [*,128] (input
 * of out_proj), dz [*,128] (gradient at norm1's input = gradient of out_proj's output); FFN block: act_out [*,256]
 * (input of linear2), dh [*,256] (gradient of linear1's output), dz [*,128] (gradient of linear2's output) -- and one
 * row [dgamma(128) | dbeta(128)] f32 of LayerNorm parameter-gradient partial sums per persistent workgroup
 * (ln_partial: ococc_window_block_partial_rows(tiles) rows; an FFN block has ceil(num_tokens / 64) tiles).
 * dx includes the residual path.  No atomics: bit-reproducible.
 * All four kernels run as persistent workgroups that keep the block's weight fragments in registers.
 * ------------------------------------------------------------------------ */
int ococc_linear_fragments_bf16(int32_t count, const void* const* src, const int64_t* rows, const int64_t* cols,
                                const int64_t* row_stride, const int64_t* col_stride, void* const* dst,
                                ococc_stream_t stream);
int64_t ococc_window_tile_plan_workspace_bytes(int64_t num_windows);
int64_t ococc_window_block_partial_rows(int64_t num_tiles);
int ococc_window_tile_plan(const int32_t* win_len, const int64_t* win_off, const int32_t* tok, int64_t num_windows,
                           int32_t tile_slots, int64_t cap_tiles, int32_t* tile_rows, int32_t* tile_span,
                           int32_t* num_tiles, void* workspace, int64_t workspace_bytes, ococc_stream_t stream);
int ococc_window_attn_block_fwd_bf16(const uint16_t* x, const uint16_t* pos, const int32_t* tile_rows,
                                     const int32_t* tile_span, int64_t num_tiles, int32_t d_model, int32_t num_heads,
                                     const uint16_t* wqkv_frag, const float* bqkv, const uint16_t* wo_frag,
                                     const float* bo, const float* ln_weight, const float* ln_bias, float eps,
                                     uint16_t* y, ococc_stream_t stream);
int ococc_window_attn_block_bwd_bf16(const uint16_t* x, const uint16_t* pos, const uint16_t* dy,
                                     const int32_t* tile_rows, const int32_t* tile_span, int64_t num_tiles,
                                     int32_t d_model, int32_t num_heads, const uint16_t* wqkv_frag, const float* bqkv,
                                     const uint16_t* wo_frag, const float* bo, const float* ln_weight, float eps,
                                     const uint16_t* wo_t_frag, const uint16_t* wqkv_t_frag, uint16_t* dx,
                                     uint16_t* dqkv, uint16_t* dz, uint16_t* attn_out, float* ln_partial,
                                     ococc_stream_t stream);
/* The same pair for a training step that keeps the attention output and the softmax's log-sum-exp (round 5): the forward
 * also writes attn_save [num_tokens, d_model] bf16 (rows of the plan's tokens) and lse_save [num_tokens, num_heads] f32; the
 * backward reads them back instead of running the attention forward again (28 % of its time, for 288 B per token) and
 * no longer writes attn_out -- the weight gradient of the out-projection reads attn_saved.  Same results as the pair
 * above (the attention forward it skips is deterministic). */
int ococc_window_attn_block_train_fwd_bf16(const uint16_t* x, const uint16_t* pos, const int32_t* tile_rows,
                                           const int32_t* tile_span, int64_t num_tiles, int32_t d_model, int32_t num_heads,
                                           const uint16_t* wqkv_frag, const float* bqkv, const uint16_t* wo_frag,
                                           const float* bo, const float* ln_weight, const float* ln_bias, float eps,
                                           uint16_t* y, uint16_t* attn_save, float* lse_save, ococc_stream_t stream);
int ococc_window_attn_block_bwd_saved_bf16(const uint16_t* x, const uint16_t* pos, const uint16_t* dy,
                                           const int32_t* tile_rows, const int32_t* tile_span, int64_t num_tiles,
                                           int32_t d_model, int32_t num_heads, const uint16_t* wqkv_frag, const float* bqkv,
                                           const uint16_t* wo_frag, const float* bo, const float* ln_weight, float eps,
                                           const uint16_t* wo_t_frag, const uint16_t* wqkv_t_frag,
                                           const uint16_t* attn_saved, const float* lse_saved, uint16_t* dx, uint16_t* dqkv,
                                           uint16_t* dz, float* ln_partial, ococc_stream_t stream);
int ococc_token_ffn_block_fwd_bf16(const uint16_t* x, int64_t num_tokens, int32_t d_model, int32_t d_ffn,
                                   const uint16_t* w1_frag, const float* b1, const uint16_t* w2_frag, const float* b2,
                                   const float* ln_weight, const float* ln_bias, float eps, int32_t act, uint16_t* y,
                                   ococc_stream_t stream);
int ococc_token_ffn_block_bwd_bf16(const uint16_t* x, const uint16_t* dy, int64_t num_tokens, int32_t d_model,
                                   int32_t d_ffn, const uint16_t* w1_frag, const float* b1, const uint16_t* w2_frag,
                                   const float* b2, const float* ln_weight, float eps, int32_t act,
                                   const uint16_t* w2_t_frag, const uint16_t* w1_t_frag, uint16_t* dx,
                                   uint16_t* act_out, uint16_t* dh, uint16_t* dz, float* ln_partial,
                                   ococc_stream_t stream);

/* Weight gradients of token-wise linears (replaces the dW / db that autograd derives for nn.Linear /
 * nn.MultiheadAttention's in_proj in the reference, sst_basic_block_v2.py:41-127): for each of `count` pairs
 *   dW_i[n][k] = sum_t G_i[t][n] X_i[t][k],  db_i[n] = sum_t G_i[t][n]
 * G_i: bf16 [num_tokens, ldg_i] (n_i columns used, n_i a multiple of 64), X_i: bf16 [num_tokens, k_i], k_i 128 or 256;
 * xadd_i (optional): a second bf16 [num_tokens, k_i] operand, rows n < add_rows_i of dW use bf16(X + xadd) (the q | k
 * rows of in_proj see x + pos, the v rows x).  The token range is cut into `slabs` (ococc_token_wgrad_slabs) and every
 * slab leaves f32 partials dw_partial_i [slabs, n_i, k_i], db_partial_i [slabs, n_i];
 * ococc_partial_rows_sum_f32 (dst[c] = sum_r src[r][c], fixed order) finishes them.  No atomics. */
int64_t ococc_token_wgrad_slabs(int64_t num_tokens);
int ococc_token_wgrad_bf16(int32_t count, const void* const* g, const int64_t* ldg, const int64_t* n,
                           const void* const* x, const void* const* xadd, const int64_t* add_rows, const int64_t* k,
                           int64_t num_tokens, int64_t slabs, void* const* dw_partial, void* const* db_partial,
                           ococc_stream_t stream);
int ococc_partial_rows_sum_f32(int32_t count, const void* const* src, const int64_t* rows, const int64_t* cols,
                               void* const* dst, ococc_stream_t stream);

/* ------------------------------------------------------------------------
 * A6, fused  one per-point layer of the SIR encoders:  y = act(LayerNorm(W x)),  x assembled per row as
 *   x = [ a (*) mul (*) colscale | b * bscale | v[inv] ]      (parts with 0 columns are absent)
 * optionally with the segment maximum of y (rows sorted by segment: inv non-decreasing) in the same launch.
 * replaces, per layer: the torch.cat / element-wise products that build the input, DynamicVFELayerV2
 * (Linear(bias=False) -> LN -> GELU, mmdet3d/models/voxel_encoders/utils.py:174-189), scatter_v2(mode='max')
 * (mmdet3d/ops/sst/sst_ops.py:150-181) and voxel_feats[unq_inv] of SIRLayer.forward
 * (mmdet3d/models/voxel_encoders/voxel_encoder.py:764-832), and the layers of rel_mlp (sst_ops.py:333-360).
 * All f32 (the reference runs these layers under force_fp32).  a [rows, lda] (ka columns used), mul [rows, ldm] or
 * null, colscale [ka] or null, b [rows, ldb] (kb columns), v [num_segments, kv], inv [rows] int32;
 * 1 <= ka + kb + kv <= 256, 1 <= n <= 144.  w_frag: ococc_point_mlp_pack_f32 of the Linear's weight [n, k]
 * (ococc_point_mlp_fragment_floats(n, k) floats); wt_frag: the same call on the transposed view (rows k, cols n).
 * ln_weight / ln_bias null: no normalisation.  act: 0 none, 1 GELU (erf), 2 ReLU.
 * fwd: y [rows, n]; seg_max [num_segments, n] or null (filled with -inf, then maxima; every segment must own a row).
 * ococc_point_mlp_segment_argmax: seg_arg [num_segments, n] = smallest row attaining the maximum.
 * bwd (recomputes the layer): dy [rows, n] or null, d_seg_max + seg_arg or null ->
 *   dz [rows, n] (gradient at the Linear's output), x_cat [rows, k] or null (the assembled input: dW = dz^T x_cat),
 *   da, dmul [rows, ka], db [rows, kb] (each or null), dv [num_segments, kv] (caller-zeroed, added to with float
 *   atomics at segment-run ends), ln_partial [ococc_point_mlp_tiles(rows), 2, n] (dgamma | dbeta partial sums).
 * ------------------------------------------------------------------------ */
int64_t ococc_point_mlp_fragment_floats(int32_t n, int32_t k);
int64_t ococc_point_mlp_tiles(int64_t rows);
/* rows per workgroup tile of the fwd / bwd launches: 0 = by input size (16 up to 4 k rows, 32 up to 131 k rows, else 64),
 * 16, 32 or 64 pinned (tests); ococc_point_mlp_tiles follows. */
int ococc_point_mlp_force_tile(int32_t tile_rows);
int ococc_point_mlp_pack_f32(const float* w, int32_t n, int32_t k, int64_t row_stride, int64_t col_stride, float* frag,
                             ococc_stream_t stream);
/* the same for up to 32 matrices in one launch (HOST tables) */
int ococc_point_mlp_pack_multi_f32(int32_t count, const void* const* w, const int32_t* n, const int32_t* k,
                                   const int64_t* row_stride, const int64_t* col_stride, void* const* frag,
                                   ococc_stream_t stream);
int ococc_point_mlp_fwd_f32(const float* a, int32_t ka, int32_t lda, const float* mul, int32_t ldm, const float* colscale,
                            const float* b, int32_t kb, int32_t ldb, float bscale, const float* v, int32_t kv,
                            const int32_t* inv, int64_t rows, const float* w_frag, int32_t n, const float* ln_weight,
                            const float* ln_bias, float eps, int32_t act, float* y, float* seg_max, int64_t num_segments,
                            ococc_stream_t stream);
int ococc_point_mlp_segment_argmax(const float* y, const float* seg_max, const int32_t* inv, int64_t rows, int32_t n,
                                   int64_t num_segments, int32_t* seg_arg, ococc_stream_t stream);
/* weight gradient of a layer from the backward call's dz [rows, n] and x_cat [rows, k]: partial [slices][n][k], one
 * product per row slice (slices = ococc_point_mlp_wgrad_slices(rows), <= 64); dW = their sum -- e.g. through
 * ococc_layernorm_param_reduce_multi with the slab read as [slices][2][n k / 2].  Replaces the torch.mm of
 * nn.Linear's backward (voxel_encoders/utils.py:147-189 DynamicVFELayerV2.linear). */
int32_t ococc_point_mlp_wgrad_slices(int64_t rows);
int ococc_point_mlp_wgrad_f32(const float* dz, const float* x_cat, int64_t rows, int32_t n, int32_t k, float* partial,
                              ococc_stream_t stream);
/* The same product for ``count`` (1..8) layers over the same rows in one launch: dz[j] [rows, n[j]], x_cat[j]
 * [rows, k[j]], partial[j] as above.  (The blocks of one SIR layer's backward.) */
int ococc_point_mlp_wgrad_multi_f32(int32_t count, const float* const* dz, const float* const* x_cat, int64_t rows,
                                    const int32_t* n, const int32_t* k, float* const* partial, ococc_stream_t stream);
int ococc_point_mlp_bwd_f32(const float* a, int32_t ka, int32_t lda, const float* mul, int32_t ldm, const float* colscale,
                            const float* b, int32_t kb, int32_t ldb, float bscale, const float* v, int32_t kv,
                            const int32_t* inv, int64_t rows, const float* w_frag, const float* wt_frag, int32_t n,
                            const float* ln_weight, const float* ln_bias, float eps, int32_t act, const float* dy,
                            const float* d_seg_max, const int32_t* seg_arg, float* dz, float* x_cat, float* da,
                            float* dmul, float* db, float* dv, float* ln_partial, ococc_stream_t stream);

/* ------------------------------------------------------------------------
 * A whole SIRLayer per call: SIRLayer.forward (mmdet3d/models/voxel_encoders/voxel_encoder.py:764-832) with LayerNorm
 * blocks and max pooling: the tile bodies of the ococc_point_mlp_* launches above plus the joins between them.
 * Blocks are numbered rel_mlp first, then vfe_layers; block b is Linear(k_b -> n[b], no bias) -> LayerNorm -> act[b]
 * (0 none, 1 GELU, 2 ReLU) with k_b = cluster_cols | n[b-1] (rel), feat_cols (+ cluster_cols) (first vfe), 2 n[b-1] (later
 * vfe).  features [rows, feat_cols] (xyz first), f_cluster [rows, cluster_cols], inv int32 non-decreasing.
 *   fwd: y_out [rows, n[last]] (+ features[:, 3:] when shortcut), groups_out [groups, sum of the vfe widths]; slab: the
 *        intermediates the backward call reads (ococc_sir_layer_fwd_floats floats).
 *   bwd: dy [rows, ld_dy >= n[last]] / d_groups [groups, ld_dgroups >= sum] (row strides in floats: column slices of wider
 *        gradients are read in place; either may be null = zero) -> dfeat [rows, feat_cols] (may be null),
 *        and per block the LayerNorm partial rows [tiles][2][n] and weight-gradient slices [slices][n][k] inside slab at
 *        the offsets ococc_sir_layer_bwd_layout reports (finish with ococc_layernorm_param_reduce_multi).  y_out: the
 *        forward's output when there is no shortcut (it is the last block's y), else unused. */
typedef struct {
  int32_t n_rel, n_vfe;
  int32_t feat_cols, cluster_cols;
  int32_t with_cluster_center, shortcut;
  float bscale;                      /* factor of the f_cluster columns appended to the first vfe block's input */
  int32_t inference;                 /* nonzero: forward only -- the arg-max rows the backward call reads are not recorded */
  const float* rel_colscale;         /* [cluster_cols] scale of the rel_mlp input, or null */
  const float* colscale;             /* [feat_cols] scale of the feature columns, or null */
  int32_t n[8];
  int32_t act[8];
  float eps[8];
  const float* w_frag[8];            /* ococc_point_mlp_pack_f32 of the weight ... */
  const float* wt_frag[8];           /* ... and of its transposed view (backward only) */
  const float* ln_weight[8];
  const float* ln_bias[8];
  const float* gate;                 /* n_rel == 0 only: a gate [rows, feat_cols] computed elsewhere (ococc_sir_rel_chains_*), or null */
  float* dgate;                      /* ... and, for the backward call, where its gradient [rows, feat_cols] goes */
} ococc_sir_layer;
int64_t ococc_sir_layer_fwd_floats(const ococc_sir_layer* layer, int64_t rows, int64_t groups);
int ococc_sir_layer_fwd_f32(const ococc_sir_layer* layer, const float* features, const float* f_cluster, const int32_t* inv,
                            int64_t rows, int64_t groups, float* slab, float* y_out, float* groups_out,
                            ococc_stream_t stream);
int ococc_sir_layer_bwd_layout(const ococc_sir_layer* layer, int64_t rows, int64_t groups, int64_t* ln_partial_off,
                               int64_t* w_partial_off, int64_t* tiles, int32_t* slices, int64_t* total_floats);
int ococc_sir_layer_bwd_f32(const ococc_sir_layer* layer, const float* features, const float* f_cluster, const int32_t* inv,
                            int64_t rows, int64_t groups, const float* fwd_slab, const float* y_out, const float* dy,
                            int64_t ld_dy, const float* d_groups, int64_t ld_dgroups, float* slab, float* dfeat,
                            ococc_stream_t stream);
/* One launch per layer and direction (+ one for the weight-gradient products): while every row tile has a workgroup of
 * its own (MI355X: up to 896 tiles of 32 rows = 28 k points) and the layer's blocks are one of the shapes of
 * csrc/sir_fused.hpp (rel_mlp of 3 blocks, 2 vfe blocks: every SIRLayer of configs[2]), each of the two calls runs the
 * blocks of a tile back to back in a persistent grid that meets at a grid-wide barrier where the segment maxima
 * (backward: their gradients) cross tiles.  Everything that crosses workgroups inside the launch goes through
 * device-scope atomics; the barrier words live in a per-(device, stream) buffer the library allocates at the first call on
 * a stream (not under stream capture).  Otherwise -- and always after ococc_sir_layer_set_fused(0) -- the same tile
 * bodies run as one launch per block: the same arithmetic, the forward results are equal to the bit.
 * ococc_sir_layer_set_fused(1) takes the one-launch form at any row count (a workgroup then walks several tiles);
 * -1 restores the default (OCOCC_SIR_FUSED=0 in the environment = 0).  A barrier wait is bounded (2 s): instead of
 * hanging the device an incomplete barrier leaves its index + 1 in a sticky device word, which ococc_sir_layer_fused_status
 * reads back (synchronises the stream; 0 = every barrier completed), AND in a host-mapped word that the library reads in
 * front of its next one-launch layer: that call (ococc_sir_layer_fwd_f32 / _bwd_f32) then returns OCOCC_ESTRANDED and every
 * later layer of the process runs as per-block launches.  ococc_sir_layer_fused_check() is the same test without a launch
 * (no synchronisation, no copy: call it wherever the host has waited for the device anyway -- end of a step, a checkpoint).
 * The persistent grid takes at most 7/8 of the workgroups that were resident TOGETHER in a one-off census launch of the same
 * kernel with the same LDS size (never more than hipOccupancyMaxActiveBlocksPerMultiprocessor promises: that API reads high
 * for some register counts) and needs all of its workgroups resident at once: meant for one process per GPU, as the
 * reference trains (tools/dist_train.sh), with ONE such layer in flight per device -- two processes or streams that put
 * such launches of ~a thousand tiles on one device at the same time can starve each other until the bounded wait gives up
 * (ococc_sir_layer_set_fused(0) / OCOCC_SIR_FUSED=0 for that case).
 * ococc_sir_layer_fused_debug(grid, timeout_ms): test hooks -- launch `grid` workgroups whatever the device holds, bound a
 * wait by `timeout_ms`; 0 restores either default, (0, 0) also clears the status words (synchronises the device). */
int ococc_sir_layer_set_fused(int32_t mode);
int ococc_sir_layer_fused_check(void);
int ococc_sir_layer_fused_debug(int32_t grid, int32_t timeout_ms);
/* The rel_mlp chains of several SIRLayers in one launch per direction (the layers of a SIR stack share the cluster
 * offsets their gates are computed from: mmdet3d/models/backbones/sir.py:67-88, ococc_bbox_head.py:237-316).
 * Chain c, block j: y = act(LayerNorm(W x)), x = f_cluster * rel_colscale (j = 0) or the block before (build_mlp,
 * sst_ops.py:333-360); the last block's y is the layer's gate (ococc_sir_layer.gate of a descriptor with n_rel = 0).
 * All chains of a call have the same depth (<= 3 blocks) and cluster_cols; <= 8 chains per call.  Block widths: <= 64
 * output channels, the last block <= 64 or 129..144 -- ococc_sir_rel_chain_fwd_floats returns -1 for a chain outside
 * that (run its rel_mlp inside its layer: n_rel > 0).
 *   fwd: slabs[c] (ococc_sir_rel_chain_fwd_floats floats) keeps the inner blocks' rows, gates[c] [rows, n last].
 *   bwd: dgates[c] [rows, n last] -> per block the LayerNorm partial rows [tiles][2][n] and weight-gradient slices
 *        [slices][n][k] inside slabs[c] at the offsets ococc_sir_rel_chain_bwd_layout reports (finish with
 *        ococc_layernorm_param_reduce_multi); f_cluster receives no gradient (the reference detaches it: voxel_encoder.py:781). */
typedef struct {
  int32_t n_blocks, cluster_cols;
  const float* rel_colscale;         /* [cluster_cols] or null */
  int32_t n[4];
  int32_t act[4];
  float eps[4];
  const float* w_frag[4];
  const float* wt_frag[4];
  const float* ln_weight[4];
  const float* ln_bias[4];
} ococc_sir_rel_chain;
int64_t ococc_sir_rel_chain_fwd_floats(const ococc_sir_rel_chain* chain, int64_t rows);
int ococc_sir_rel_chain_bwd_layout(const ococc_sir_rel_chain* chain, int64_t rows, int64_t* ln_partial_off, int64_t* w_partial_off,
                                   int64_t* tiles, int32_t* slices, int64_t* total_floats);
int ococc_sir_rel_chains_fwd_f32(int32_t count, const ococc_sir_rel_chain* chains, const float* f_cluster, int64_t rows,
                                 float* const* slabs, float* const* gates, ococc_stream_t stream);
int ococc_sir_rel_chains_bwd_f32(int32_t count, const ococc_sir_rel_chain* chains, const float* f_cluster, int64_t rows,
                                 const float* const* fwd_slabs, const float* const* gates, const float* const* dgates,
                                 float* const* slabs, ococc_stream_t stream);
int ococc_sir_layer_fused_status(ococc_stream_t stream, int32_t* status);

/* ------------------------------------------------------------------------
 * A9 / A10  f32 matrix products on the bf16 matrix cores at f32-level accuracy: the operand split.
 * Replaces the f32 GEMMs behind nn.Linear / nn.MultiheadAttention of the temporal transformer and the RoI-level MLPs
 * (mmdet3d/models/occ/layers.py:35-87, mmdet3d/models/roi_heads/bbox_heads/ococc_bbox_head.py:116-193,849-908) from a few
 * hundred rows on: x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi), and
 *   x w ~ hi_x hi_w + hi_x lo_w + lo_x hi_w   (relative error 4.5e-6 against f64; the f32 GEMM: 7e-7; bf16 operands: 2.3e-3)
 * is ONE bf16 GEMM with f32 accumulation over a three times longer contraction.  ococc_split3_bf16 makes the three-part
 * operand of an f32 matrix src [rows, cols] (row stride ld_src floats) in one pass, in either or both forms:
 *   cat_cols   [rows, 3 cols] bf16: the parts side by side (the matrix's ROWS are contracted against another's rows)
 *   stack_rows [3 rows, cols] bf16: the parts stacked (its COLUMNS stay, the three row blocks are contracted)
 * pattern 0 = (hi, hi, lo), 1 = (hi, lo, hi): one operand of a product takes 0, the other 1.  cols % 8 == 0. */
int ococc_split3_bf16(const float* src, int64_t rows, int64_t cols, int64_t ld_src, uint16_t* cat_cols, int32_t cat_pattern,
                      uint16_t* stack_rows, int32_t stack_pattern, ococc_stream_t stream);

/* ------------------------------------------------------------------------
 * A9  the attention core of the temporal transformer, one launch per direction (csrc/causal_attn.hip).
 * Replaces, inside nn.MultiheadAttention as SimpleEncoderLayer calls it (mmdet3d/models/occ/layers.py:35-87; callers
 * ococc_bbox_head.py:849-995), the chain  (q / sqrt(D)) k^T -> masked_fill(attn_mask, key_padding_mask) -> softmax ->
 * dropout -> @ v  and its backward (ten launches on [B H, L, S] tensors at L = 32).
 * q, k, v: f32 token-major rows t = l * batch + b (the reference's [L, B, E] layout flattened), head h in columns
 * h * head_dim .. + head_dim - 1, row strides ldq / ldk / ldv floats (q and k may be column slices of one projection).
 * attn_mask [L, S] / key_padding_mask [batch, S]: bytes, 1 = masked, or null.  dropout_p in [0, 1): the keep mask is a
 * counter-based hash of (seed, element), regenerated by the backward call from the same seed; seed_dev non-null: the seed
 * is read from device memory (inside a captured graph every replay then draws its own).  probs [batch * heads, L, S]: the
 * probabilities before dropout, written by the forward call and read by the backward call.  out [L * batch, ldo].
 * L, S <= 256, head_dim % 4 == 0 and <= 384 (OCOCC_EUNSUPPORTED otherwise).  No atomics; deterministic. */
int ococc_temporal_attention_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                     const uint8_t* attn_mask, const uint8_t* key_padding_mask, int32_t batch, int32_t heads,
                                     int32_t L, int32_t S, int32_t head_dim, float scale, float dropout_p, uint64_t seed,
                                     const uint64_t* seed_dev, float* probs, float* out, int64_t ldo, ococc_stream_t stream);
int ococc_temporal_attention_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                     int32_t batch, int32_t heads, int32_t L, int32_t S, int32_t head_dim, float scale,
                                     float dropout_p, uint64_t seed, const uint64_t* seed_dev, const float* probs,
                                     const float* out, int64_t ldo, const float* d_out, int64_t lddo, float* dq, int64_t lddq,
                                     float* dk, int64_t lddk, float* dv, int64_t lddv, ococc_stream_t stream);

/* ------------------------------------------------------------------------
 * A11 / A10, fused  the occupancy decoder's per-query MLP, one launch per layer (or one for the whole MLP):
 *   y = dropout(act(LayerNorm(x W^T + bias + add_rows[add_index]))),  optionally  head = y . head_weight + head_bias
 * replaces OccDecoder.forward's conv_occ (mmdet3d/models/occ/occ_base.py:99-153): build_mlp's
 * Sequential(Linear(bias=False), LN, GELU, Dropout) blocks 1596 -> 512 -> 1024 -> 1024 and the Linear(1024 -> 1) head
 * (mmdet3d/ops/sst/sst_ops.py:333-360), and PosEncode.forward (occ_base.py:33-57).
 * bf16 operands, f32 accumulation, f32 LayerNorm statistics over the f32 sums, bf16 activations.
 * ococc_pos_encode_bf16: xyz f32 [rows, 3] -> out bf16 [rows, ld], columns [sin(pi 2^l x^) | cos(pi 2^l x^)] in the
 *   reference's [2L][3] order, x^ = x normalised to [-1, 1] by bound = {lo xyz, hi xyz} (HOST pointer; null: x^ = x),
 *   columns 6L .. ld - 1 zero.
 * ococc_linear_fragments32_bf16: `count` (<= 16) f32 matrices S_i [rows_i, cols_i] (any strides; rows a multiple of
 *   32) -> bf16 operand fragments of v_mfma_f32_32x32x16_bf16, columns zero-padded to padded_cols_i (multiple of 16):
 *   dst_i[rb][cs][lane][j] = S_i[32 rb + (lane & 31)][16 cs + 8 (lane >> 5) + j].  The tables are HOST arrays.
 * ococc_mlp_layer_fwd_bf16: x bf16 [rows, k], k a multiple of 64 up to 1024; w_frag the fragments of W [n, k],
 *   n 512 or 1024; bias [n], add_rows f32 [*, n] with add_index int32 [rows], ln_weight / ln_bias [n], head_weight [n] with
 *   head_bias [1] (device): each optional.  act: 0 none, 1 GELU (erf).  drop_threshold > 0: the counter-based keep mask of
 *   ococc_layernorm_act_dropout_fwd_bf16 (same threshold and seed -> same mask).  y bf16 [rows, n] or null when only the head is wanted;
 *   head_out f32 [rows] (the head reads the bf16-rounded activation, as the next Linear would).
 * ococc_occ_mlp_fwd_bf16: the reference's decoder widths, 60 (padded to 64) -> 512 -> 1024 -> 1024 -> 1, in ONE launch;
 *   a 64-row tile's activations stay in LDS between the layers.  pe bf16 [rows, 64] (ococc_pos_encode_bf16), add_rows
 *   f32 [*, 512] with add_index [rows] (the per-RoI half of the first layer); w_frag / ln_weight / ln_bias: HOST tables
 *   of 3 device pointers (fragments of [512, 64], [1024, 512], [1024, 1024]; f32 [n]); GELU after every LayerNorm;
 *   head_weight f32 [1024], head_bias [1] or null; dropout_seeds: HOST array of 3 (read when drop_threshold > 0), the
 *   masks of ococc_mlp_layer_fwd_bf16 called per layer with the same seeds.  out f32 [rows];  y0_out bf16 [rows, 512]
 *   / y1_out bf16 [rows, 1024]: optional copies of the hidden activations (what a backward pass starts from).
 *   Bit-identical to three ococc_mlp_layer_fwd_bf16 calls.
 * ------------------------------------------------------------------------ */
int ococc_pos_encode_bf16(const float* xyz, int64_t rows, const float* bound, int32_t num_freqs, uint16_t* out, int32_t ld,
                          ococc_stream_t stream);
int ococc_linear_fragments32_bf16(int32_t count, const void* const* src, const int64_t* rows, const int64_t* cols,
                                  const int64_t* padded_cols, const int64_t* row_stride, const int64_t* col_stride,
                                  void* const* dst, ococc_stream_t stream);
int ococc_mlp_layer_fwd_bf16(const uint16_t* x, int64_t rows, int32_t k, const uint16_t* w_frag, int32_t n,
                             const float* bias, const float* add_rows, const int32_t* add_index, const float* ln_weight,
                             const float* ln_bias, float eps, int32_t act, uint32_t drop_threshold, uint64_t dropout_seed,
                             uint16_t* y, const float* head_weight, const float* head_bias, float* head_out,
                             ococc_stream_t stream);
int ococc_occ_mlp_fwd_bf16(const uint16_t* pe, int64_t rows, const float* add_rows, const int32_t* add_index,
                           const void* const* w_frag, const void* const* ln_weight, const void* const* ln_bias, float eps,
                           const float* head_weight, const float* head_bias, uint32_t drop_threshold,
                           const uint64_t* dropout_seeds, uint16_t* y0_out, uint16_t* y1_out, float* out,
                           ococc_stream_t stream);
/* the same launch in a training step: per layer l it also writes the LayerNorm input z_out[l] (bf16 [rows, n_l]; the
 * LayerNorm works on these rounded values), its row statistics stats_out[l] (f32 [rows, 2]: mean, rstd) and the activation
 * y_out[l] (bf16, after GELU and dropout) -- what the backward pass of OccDecoder.forward's conv_occ reads
 * (ococc_layernorm_act_bwd / _dropout_bwd_bf16 with the same (threshold, seed) per layer). */
int ococc_occ_mlp_train_fwd_bf16(const uint16_t* pe, int64_t rows, const float* add_rows, const int32_t* add_index,
                                 const void* const* w_frag, const void* const* ln_weight, const void* const* ln_bias,
                                 float eps, const float* head_weight, const float* head_bias, uint32_t drop_threshold,
                                 const uint64_t* dropout_seeds, void* const* z_out, void* const* y_out,
                                 void* const* stats_out, float* out, ococc_stream_t stream);
/* (z_out, y_out and stats_out may be null, all three: the forward of the recomputing backward below -- the same numbers,
 * nothing kept but the logits.)
 *
 * The BACKWARD pass of that forward in one launch, with recompute -- replaces what autograd derives for
 * OccDecoder.forward's conv_occ (mmdet3d/models/occ/occ_base.py:99-153; the Sequential(Linear, LN, GELU, Dropout) blocks
 * of mmdet3d/ops/sst/sst_ops.py:333-360): per 64-row tile the three layers are run forward again from the positional
 * encodings (same numbers as the training forward), then d logit is taken back through the head, the three
 * LayerNorm / GELU / dropout backward steps and the two input-gradient GEMMs (w_t_frag: fragments of W1^T [512, 1024] and
 * W2^T [1024, 1024], ococc_linear_fragments32_bf16 on the transposed views; a third entry is ignored).  Leaves what the
 * weight gradients contract over the rows -- y_out[0] = y0 [rows, 512], y_out[1] = y1 [rows, 1024], dz_out[l] = d z_l
 * (bf16; d z_0 is also the gradient of the gathered per-RoI rows add_rows[add_index]) -- and per workgroup the sums
 * [d gamma_0 | d beta_0 | d gamma_1 | d beta_1 | d gamma_2 | d beta_2 | d head_weight] in partials
 * [ococc_occ_mlp_bwd_workgroups(rows)][ococc_occ_mlp_bwd_partial_cols()] f32, which must be ZERO on entry (a column sum
 * finishes them).  scratch: ococc_occ_mlp_bwd_scratch_bytes(rows), 16-byte aligned, contents unspecified.
 *
 * Without recompute: z_parked[l] / stats[l] non-null -- what ococc_occ_mlp_train_fwd_bf16 left when its y_out[2] was null
 * (z_out[l] then hold the LayerNorm inputs in the kernel's own per-tile lane order, bf16 [ceil(rows / 64) * 64, n_l]
 * elements; y_out[0], y_out[1] row-major as always; y2 is not kept).  The kernel then starts at d logit: pe, add_rows,
 * add_index, w_frag, y_out and scratch are ignored. */
int ococc_occ_mlp_bwd_bf16(const uint16_t* pe, int64_t rows, const float* add_rows, const int32_t* add_index,
                           const void* const* w_frag, const void* const* w_t_frag, const void* const* ln_weight,
                           const void* const* ln_bias, float eps, const float* head_weight, const float* dlogit,
                           uint32_t drop_threshold, const uint64_t* dropout_seeds, const void* const* z_parked,
                           const void* const* stats, void* const* y_out, void* const* dz_out, float* partials,
                           void* scratch, int64_t scratch_bytes, ococc_stream_t stream);
int64_t ococc_occ_mlp_bwd_workgroups(int64_t rows);
int64_t ococc_occ_mlp_bwd_partial_cols(void);
int64_t ococc_occ_mlp_bwd_scratch_bytes(int64_t rows);

/* ------------------------------------------------------------------------
 * B6, element-wise halves of the SST input layer, one launch each (the mirror ran 20-50 torch operators per call):
 * ococc_sst_window_coors_i64: get_window_coors (mmdet3d/ops/sst/sst_ops.py:266-313) for BOTH window shifts.
 *   coors int64 [n, 4] (b, z, y, x); shapes as (x, y, z) HOST triples (2-D windows: pass window z = sparse z);
 *   win_ids int64 [2, n] (shift 0 = unshifted, 1 = shifted by half a window), coors_in_win int64 [2, n, 3] (z, y, x).
 * ococc_sst_drop_level_i64: the level / keep decision of SSTInputLayerV2.drop_single_shift
 *   (mmdet3d/models/middle_encoders/sst_input_layer_v2.py:128-148) from one group-rank pass: window population =
 *   counts[conti[i]]; the LAST table row whose [lower, upper) holds it gives level_ids[r] and max_tokens[r] (none: -1,
 *   0); keep[i] = inner[i] < max_tokens.  Table rows are HOST arrays (<= 8).
 * ococc_sst_pos_embed: get_pos_embed (sst_input_layer_v2.py:239-305) in flat token order: out [n, feat_dim] f32 / bf16,
 *   columns [x | y | z] blocks of pos_length (sin at even, cos at odd columns of e = p / inv_freq), zero padded;
 *   inv_freq: DEVICE f32 [pos_length], the table the reference builds with torch.pow.
 * ------------------------------------------------------------------------ */
int ococc_sst_window_coors_i64(const int64_t* coors, int64_t n, const int32_t* sparse_shape_xyz,
                               const int32_t* window_shape_xyz, int64_t* win_ids, int64_t* coors_in_win,
                               ococc_stream_t stream);
int ococc_sst_drop_level_i64(const int32_t* conti, const int32_t* inner, const int32_t* counts, int64_t n,
                             int32_t num_levels, const int64_t* lower, const int64_t* upper, const int64_t* max_tokens,
                             const int64_t* level_ids, uint8_t* keep, int64_t* level, ococc_stream_t stream);
int ococc_sst_pos_embed(const int64_t* coors_in_win, int64_t n, const int32_t* window_shape_xyz, int32_t ndim,
                        int32_t normalize_pos, const float* inv_freq, int32_t pos_length, int32_t feat_dim, void* out,
                        int32_t out_dtype, ococc_stream_t stream);

/* f32 <-> bf16 row casts (round to nearest even) */
int ococc_cast_f32_to_bf16(const float* src, uint16_t* dst, int64_t count, ococc_stream_t stream);
int ococc_cast_bf16_to_f32(const uint16_t* src, float* dst, int64_t count, ococc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused multi-tensor AdamW step, torch.optim.AdamW semantics (amsgrad / maximize off) -- the
 * optimizer the reference configures at configs/_base_/schedules/cosine_2x.py:2-8 (lr override
 * configs/ococc/ococcnet.py:468-470).  params / grads / exp_avg / exp_avg_sq are HOST arrays of
 * num_tensors (<= 48) device pointers to contiguous f32 tensors of numel[i] elements.  `step` is a
 * DEVICE float holding the number of steps taken so far: the kernel uses step + 1 for the bias
 * corrections; with bump_step = 1 a one-thread kernel queued behind it stores step + 1 (pass 0 on all
 * but the last call when one optimizer step needs several calls), so the sequence can be replayed from
 * a captured HIP graph.  bump_step = 2: `step` points to {float count; uint32 ticket}, ticket zero before
 * the first call; launches of up to 2048 workgroups (2 M elements) then store count + 1 from their last
 * workgroup instead of a second launch (larger ones fall back to the one-thread kernel).
 * ------------------------------------------------------------------------------------------- */
int ococc_adamw_f32(int32_t num_tensors, void* const* params, const void* const* grads,
                    void* const* exp_avg, void* const* exp_avg_sq, const int64_t* numel, float lr,
                    float beta1, float beta2, float eps, float weight_decay, float* step,
                    int32_t bump_step, ococc_stream_t stream);
/* The same update with the learning rate read from device memory at run time (one float): a learning-rate schedule
 * (the reference's cyclic policy, configs/_base_/schedules/cosine_2x.py:10-15, applied per iteration by mmcv's
 * CyclicLrUpdaterHook) then takes effect between replays of a captured HIP graph, where a scalar launch argument
 * would stay frozen at its capture-time value. */
int ococc_adamw_lr_dev_f32(int32_t num_tensors, void* const* params, const void* const* grads,
                           void* const* exp_avg, void* const* exp_avg_sq, const int64_t* numel,
                           const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                           float* step, int32_t bump_step, ococc_stream_t stream);
/* The same update that ALSO refreshes the bf16 kernel operands of convolution weights it has just written (what
 * ococc_weight_prepare_multi_bf16 would compute at the start of the next step: one launch less per step).  lr_dev null:
 * lr is used.  Operand o belongs to tensor operand_tensor[o] (an index into params, a contiguous f32 [kvol, cin, cout]),
 * is written in layout operand_mode[o] (the modes of ococc_weight_prepare_bf16, + 4 = fragment-major) to operand_dst[o]
 * (bf16, kvol * cin * cout elements); at most 8 operands per call.  All tables are HOST arrays. */
int ococc_adamw_operands_f32(int32_t num_tensors, void* const* params, const void* const* grads, void* const* exp_avg,
                             void* const* exp_avg_sq, const int64_t* numel, float lr, const float* lr_dev, float beta1,
                             float beta2, float eps, float weight_decay, float* step, int32_t bump_step,
                             int32_t num_operands, const int32_t* operand_tensor, const int32_t* operand_mode,
                             const int32_t* operand_kvol, const int32_t* operand_cin, const int32_t* operand_cout,
                             void* const* operand_dst, ococc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Stream timers (HIP events) for the measurement harness (bench.py roofline line).  No reference
 * counterpart.  in_graph != 0 records with hipEventRecordExternal, i.e. as an event-record node
 * of the HIP graph being captured on `stream`, so that a kernel inside a replayed graph can be
 * timed; the elapsed time is read after the stream has been synchronised.
 * ------------------------------------------------------------------------------------------- */
int ococc_timer_create(void** timer);
int ococc_timer_record(void* timer, int32_t in_graph, ococc_stream_t stream);
int ococc_timer_elapsed_ms(void* start, void* stop, float* ms);
int ococc_timer_destroy(void* timer);

#ifdef __cplusplus
}
#endif
#endif /* OCOCC_HIP_H_ */
