// Unique rows of bounded integer coordinates, sorted, with inverse map and
// counts -- what the reference obtains from at::unique_dim
// (mmdet3d/ops/voxel/src/scatter_points_cuda.cu:199-210) and torch.unique
// (mmdet3d/ops/sst/sst_ops.py:155-158).
//
// MI355X design: the coordinate space of an object grid is small and bounded
// (batch * D * H * W cells), so instead of a comparison sort we keep ONE BIT per
// cell (0.5 MB for 64 grids of 40^3: L2 resident), mark occupied cells with
// integer atomics, and turn the bitmap into ranks with a popcount prefix sum.
// rank(cell) = prefix[cell/32] + popc(bits below) is exactly the position of
// the row in lexicographic order.  No float atomics, deterministic output.
// Traffic per point: 4*ndim B read + 4 B inv written + two 4 B bitmap hits.
#include "common.hpp"
#include "scan.hpp"

namespace {

struct Dims {
  int32_t d[4];
  int32_t ndim;
};

__global__ void __launch_bounds__(256)
mark_cells_kernel(const int32_t* __restrict__ coors, int64_t n, Dims dims,
                  uint32_t* __restrict__ bitmap, int32_t* __restrict__ cell_of,
                  int32_t* __restrict__ status) {
  const int64_t n_up = (n + 63) & ~(int64_t)63;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_up;
       i += (int64_t)gridDim.x * blockDim.x) {
    const bool in_range = i < n;
    const int32_t* c = coors + (in_range ? i : 0) * dims.ndim;
    int64_t cell = 0;
    bool neg = !in_range, over = false;
    for (int k = 0; k < dims.ndim; ++k) {
      int32_t v = c[k];
      neg |= v < 0;
      over |= v >= dims.d[k];
      cell = cell * dims.d[k] + v;
    }
    if (over && !neg) {
      *status = 1;  // benign race: every writer stores 1
      neg = true;
    }
    if (!in_range) {
    } else if (neg) {
      cell_of[i] = -1;
    } else {
      cell_of[i] = (int32_t)cell;
    }
    // Keys repeat, and in runs -- the pooled points arrive sorted by RoI, so the 64 lanes of a wave name one or two cells:
    // 33 k points on 512 keys were 33 k atomics on 16 words, ~11 ns apiece (373 us).  Only the first lane of a run inside
    // the wave issues the atomic (every lane of the wave takes part in the shuffle: the loop bound is rounded up to whole waves).
    const int32_t mine = neg ? -1 : (int32_t)cell;
    const int32_t left = __shfl_up(mine, 1, 64);
    if (!neg && ((threadIdx.x & 63) == 0 || left != mine)) {
      const uint32_t bit = 1u << (cell & 31);
      if (!(__builtin_nontemporal_load(bitmap + (cell >> 5)) & bit)) atomicOr(bitmap + (cell >> 5), bit);
    }
  }
}

__global__ void __launch_bounds__(256)
assign_rank_kernel(int64_t n, const uint32_t* __restrict__ bitmap,
                   const uint32_t* __restrict__ prefix, int32_t* __restrict__ inv,
                   int32_t* __restrict__ counts, int64_t out_capacity) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int32_t cell = inv[i];
    if (cell < 0) continue;
    uint32_t w = bitmap[cell >> 5];
    uint32_t r = prefix[cell >> 5] + __popc(w & ((1u << (cell & 31)) - 1u));
    inv[i] = (int32_t)r;
    if (counts && (int64_t)r < out_capacity) atomicAdd(counts + r, 1);
  }
}

__global__ void __launch_bounds__(256)
emit_coors_kernel(int64_t words, const uint32_t* __restrict__ bitmap,
                  const uint32_t* __restrict__ prefix, Dims dims, int32_t* __restrict__ out_coors,
                  int64_t out_capacity) {
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words;
       w += (int64_t)gridDim.x * blockDim.x) {
    uint32_t bits = bitmap[w];
    if (!bits) continue;
    int64_t r = prefix[w];
    while (bits) {
      int b = __ffs(bits) - 1;
      bits &= bits - 1;
      if (r < out_capacity) {
        int64_t cell = w * 32 + b;
        int32_t* o = out_coors + r * dims.ndim;
        for (int k = dims.ndim - 1; k >= 0; --k) {
          o[k] = (int32_t)(cell % dims.d[k]);
          cell /= dims.d[k];
        }
      }
      ++r;
    }
  }
}

// ---------------------------------------------------------------- fused voxelise + scatter-mean
// Front end of the per-object occupancy encoder in one pass structure: points -> cells (the
// arithmetic of voxelize.hip), cell bitmap -> ranks (as above), and the DynamicScatter mean of the
// point features, without the intermediate [N,3] / [N,4] coordinate tensors and without zero-filling
// or atomically accumulating the voxel features: the FIRST point to reach a cell (the one whose
// atomicOr found the bit clear) plain-stores its features into the voxel's row, which for object
// grids is nearly every point (128 k random points in 64 x 40^3 cells: 98.4 % are alone in their
// cell).  Only the later arrivals ("dups") go through float atomics, in a second pass, and only the
// rows they touch are divided by their count in a third.
struct VoxGeom {
  float vx, vy, vz, xmin, ymin, zmin;
  int32_t gx, gy, gz, batch;
};

// code[i] = 2 * cell + dup (dup: the cell's bit was already set), or -1 / -2 for a dropped point
__global__ void __launch_bounds__(256)
voxel_mark_kernel(const float* __restrict__ points, int nfeat, const int32_t* __restrict__ batch_idx,
                  int64_t n, VoxGeom g, uint32_t* __restrict__ bitmap, int32_t* __restrict__ code_of,
                  int32_t* __restrict__ status) {
  // nothing in this launch reads it: zeroed here for the launch behind it (saves a fill)
  if (blockIdx.x == 0 && threadIdx.x == 0) *status = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = points + i * nfeat;
    // voxelization_cpu.cpp:8-41: floor((p - min) / voxel) in float, clamped to the grid
    int cx = (int)floorf((p[0] - g.xmin) / g.vx);
    int cy = (int)floorf((p[1] - g.ymin) / g.vy);
    int cz = (int)floorf((p[2] - g.zmin) / g.vz);
    cx = cx < 0 ? 0 : (cx >= g.gx ? g.gx - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= g.gy ? g.gy - 1 : cy);
    cz = cz < 0 ? 0 : (cz >= g.gz ? g.gz - 1 : cz);
    const int32_t b = batch_idx[i];
    int32_t code = b >= g.batch ? -2 : -1;  // -2: outside the declared batch, reported by the next launch
    if (b >= 0 && b < g.batch) {
      const int32_t cell = ((b * g.gz + cz) * g.gy + cy) * g.gx + cx;
      const uint32_t bit = 1u << (cell & 31);
      const uint32_t old = atomicOr(bitmap + (cell >> 5), bit);
      code = cell * 2 + ((old & bit) ? 1 : 0);
    }
    code_of[i] = code;
  }
}

__device__ __forceinline__ void store_feat_piece(const float* __restrict__ src, float* __restrict__ dst_f32,
                                                 uint16_t* __restrict__ dst_bf16, int c, int piece, float scale) {
  // 4 channels starting at 4 * piece (the last piece may be shorter)
  const int c0 = piece * 4;
  if (c0 + 4 <= c && (c & 3) == 0) {
    float4 v = *(const float4*)(src + c0);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    if (dst_f32) *(float4*)(dst_f32 + c0) = v;
    if (dst_bf16) {
      uint2 q;
      q.x = (uint32_t)ococc_f32_to_bf16(v.x) | ((uint32_t)ococc_f32_to_bf16(v.y) << 16);
      q.y = (uint32_t)ococc_f32_to_bf16(v.z) | ((uint32_t)ococc_f32_to_bf16(v.w) << 16);
      *(uint2*)(dst_bf16 + c0) = q;
    }
  } else {
    for (int ch = c0; ch < c0 + 4 && ch < c; ++ch) {
      const float v = src[ch] * scale;
      if (dst_f32) dst_f32[ch] = v;
      if (dst_bf16) dst_bf16[ch] = ococc_f32_to_bf16(v);
    }
  }
}

// Three roles in one launch, told apart by the block index:
//   A  [0, blocks_a)         thread = (point, 16-byte piece): rank of the point's cell -> inv; the
//                            cell's first arrival copies its features (dups wait for the next launch)
//   B  [blocks_a, +blocks_b) thread = bitmap word: coordinates (b,z,y,x) and count 1 of its voxels
//   C  the rest              thread = output row past the voxel count: -1 coordinates, zero features
__global__ void __launch_bounds__(256)
voxel_emit_kernel(int64_t n, int c, int pieces, const float* __restrict__ feats,
                  const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ prefix, int64_t words,
                  VoxGeom g, int blocks_a, int blocks_b, const int32_t* __restrict__ code_of,
                  int32_t* __restrict__ inv, int32_t* __restrict__ out_coors, int32_t* __restrict__ counts, float* __restrict__ out_f32,
                  uint16_t* __restrict__ out_bf16, int64_t cap, const int32_t* __restrict__ num_voxels,
                  int32_t* __restrict__ status) {
  if ((int)blockIdx.x < blocks_a) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t / pieces;
    const int piece = (int)(t - i * pieces);
    if (i >= n) return;
    const int32_t code = code_of[i];
    if (code < 0) {
      if (piece == 0) {
        inv[i] = -1;
        if (code == -2) *status = 1;  // benign race: every writer stores 1
      }
      return;
    }
    const int32_t cell = code >> 1;
    const uint32_t w = bitmap[cell >> 5];
    const int64_t r = (int64_t)prefix[cell >> 5] + __popc(w & ((1u << (cell & 31)) - 1u));
    if (!(code & 1) && r < cap) {
      store_feat_piece(feats + i * c, out_f32 ? out_f32 + r * c : nullptr, out_bf16 ? out_bf16 + r * c : nullptr, c,
                       piece, 1.f);
    }
    if (piece == 0) inv[i] = (int32_t)r;
  } else if ((int)blockIdx.x < blocks_a + blocks_b) {
    const int64_t wi = (int64_t)(blockIdx.x - blocks_a) * blockDim.x + threadIdx.x;
    if (wi >= words) return;
    uint32_t bits = bitmap[wi];
    int64_t r = prefix[wi];
    while (bits) {
      const int bb = __ffs(bits) - 1;
      bits &= bits - 1;
      if (r < cap) {
        int64_t cell = wi * 32 + bb;
        int32_t* o = out_coors + r * 4;
        o[3] = (int32_t)(cell % g.gx); cell /= g.gx;
        o[2] = (int32_t)(cell % g.gy); cell /= g.gy;
        o[1] = (int32_t)(cell % g.gz); cell /= g.gz;
        o[0] = (int32_t)cell;
        counts[r] = 1;
      }
      ++r;
    }
  } else {
    const int64_t t = (int64_t)(blockIdx.x - blocks_a - blocks_b) * blockDim.x + threadIdx.x;
    const int64_t r = t / pieces;
    const int piece = (int)(t - r * pieces);
    if (r >= cap || r < (int64_t)*num_voxels) return;
    if (piece == 0) {
      *(int4*)(out_coors + r * 4) = make_int4(-1, -1, -1, -1);
      counts[r] = 0;
    }
    for (int ch = piece * 4; ch < piece * 4 + 4 && ch < c; ++ch) {
      if (out_f32) out_f32[r * c + ch] = 0.f;
      if (out_bf16) out_bf16[r * c + ch] = 0;
    }
  }
}

// Later arrivals (thread = point; they are few): add the features to the row their cell's first
// arrival wrote and count them; the one that takes the count from 1 to 2 marks itself (code -3) as the
// row's finaliser.  (A list of dups behind ONE atomic counter was measured at 23 us for 2 k dups.)
__global__ void __launch_bounds__(256)
voxel_dup_add_kernel(int64_t n, int c, const float* __restrict__ feats, const int32_t* __restrict__ inv,
                     int32_t* __restrict__ code_of, int32_t* __restrict__ counts, float* __restrict__ sums,
                     int64_t cap) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t code = code_of[i];
    if (code < 0 || !(code & 1)) continue;
    const int32_t r = inv[i];
    if (r >= cap) continue;
    for (int ch = 0; ch < c; ++ch) atomicAdd(sums + (int64_t)r * c + ch, feats[i * c + ch]);
    if (atomicAdd(counts + r, 1) == 1) code_of[i] = -3;
  }
}

// rows with more than one point: sum -> mean (scatter_points_cuda.cu:225-241 divides by the point count);
// thread = (point, 16-byte piece), only the finalisers act
__global__ void __launch_bounds__(256)
voxel_dup_mean_kernel(int64_t n, int c, int pieces, const int32_t* __restrict__ code_of,
                      const int32_t* __restrict__ inv, const int32_t* __restrict__ counts,
                      float* __restrict__ sums, uint16_t* __restrict__ out_bf16) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * pieces; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / pieces;
    if (code_of[i] != -3) continue;
    const int piece = (int)(t - i * pieces);
    const int64_t r = inv[i];
    const float cnt = (float)counts[r];
    for (int ch = piece * 4; ch < piece * 4 + 4 && ch < c; ++ch) {
      const float v = sums[r * c + ch] / cnt;
      sums[r * c + ch] = v;
      if (out_bf16) out_bf16[r * c + ch] = ococc_f32_to_bf16(v);
    }
  }
}

struct Layout {
  int64_t words, off_bitmap, off_prefix, off_scratch, total;
};

inline bool make_layout(int32_t ndim, const int32_t* dims, Layout* L) {
  if (ndim < 1 || ndim > 4) return false;
  int64_t cells = 1;
  for (int i = 0; i < ndim; ++i) {
    if (dims[i] < 1) return false;
    cells *= dims[i];
    if (cells > 0x7fffffffLL) return false;
  }
  L->words = (cells + 31) / 32;
  L->off_bitmap = 0;
  L->off_prefix = ococc_align_up(L->words * 4, 256);
  L->off_scratch = L->off_prefix + ococc_align_up(L->words * 4, 256);
  L->total = L->off_scratch + ococc_align_up(ococc_scan::scratch_words(L->words, 1) * 4, 256);
  return true;
}

}  // namespace

extern "C" int64_t ococc_grid_unique_workspace_bytes(int32_t ndim, const int32_t host_dims[4]) {
  Layout L;
  if (!host_dims || !make_layout(ndim, host_dims, &L)) return -1;
  return L.total;
}

extern "C" int ococc_grid_unique_workspace_layout(int32_t ndim, const int32_t host_dims[4],
                                                  int64_t* bitmap_offset, int64_t* prefix_offset) {
  Layout L;
  OCOCC_REQUIRE(host_dims && make_layout(ndim, host_dims, &L), "ndim must be 1..4 and prod(dims) < 2^31");
  OCOCC_REQUIRE(bitmap_offset && prefix_offset, "null output");
  *bitmap_offset = L.off_bitmap;
  *prefix_offset = L.off_prefix;
  return OCOCC_OK;
}

extern "C" int ococc_grid_unique_i32(const int32_t* coors, int64_t n, int32_t ndim,
                                     const int32_t host_dims[4], int32_t* out_coors,
                                     int64_t out_capacity, int32_t* inv, int32_t* counts,
                                     int32_t* num_unique, int32_t* status, void* workspace,
                                     int64_t workspace_bytes, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Layout L;
  OCOCC_REQUIRE(host_dims && make_layout(ndim, host_dims, &L),
                "ndim must be 1..4 and prod(dims) < 2^31");
  OCOCC_REQUIRE(n >= 0 && out_capacity >= 0, "negative size");
  OCOCC_REQUIRE(num_unique && status, "num_unique/status must be device pointers");
  OCOCC_REQUIRE(workspace && workspace_bytes >= L.total, "workspace too small");
  // status is a flag the kernels only ever set; num_unique is overwritten by the scan below
  if (status == num_unique + 1) {
    OCOCC_HIP(hipMemsetAsync(num_unique, 0, 2 * sizeof(int32_t), stream));
  } else {
    OCOCC_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), stream));
    if (n == 0) OCOCC_HIP(hipMemsetAsync(num_unique, 0, sizeof(int32_t), stream));
  }
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(coors && inv, "null coors/inv");
  char* ws = (char*)workspace;
  uint32_t* bitmap = (uint32_t*)(ws + L.off_bitmap);
  uint32_t* prefix = (uint32_t*)(ws + L.off_prefix);
  uint32_t* scratch = (uint32_t*)(ws + L.off_scratch);
  Dims dims;
  dims.ndim = ndim;
  for (int i = 0; i < 4; ++i) dims.d[i] = i < ndim ? host_dims[i] : 1;

  OCOCC_HIP(hipMemsetAsync(bitmap, 0, L.words * 4, stream));
  if (counts && out_capacity > 0)
    OCOCC_HIP(hipMemsetAsync(counts, 0, out_capacity * sizeof(int32_t), stream));
  hipLaunchKernelGGL(mark_cells_kernel, dim3(ococc_grid_1d(n, 256)), dim3(256), 0, stream, coors, n,
                     dims, bitmap, inv, status);
  OCOCC_CHECK_LAUNCH();
  OCOCC_HIP(ococc_scan::exclusive_scan<ococc_scan::POPC>(bitmap, L.words, L.words, 1, prefix,
                                                         L.words, scratch, (uint32_t*)num_unique,
                                                         stream));
  hipLaunchKernelGGL(assign_rank_kernel, dim3(ococc_grid_1d(n, 256)), dim3(256), 0, stream, n,
                     bitmap, prefix, inv, counts, out_capacity);
  OCOCC_CHECK_LAUNCH();
  if (out_coors && out_capacity > 0) {
    hipLaunchKernelGGL(emit_coors_kernel, dim3(ococc_grid_1d(L.words, 256)), dim3(256), 0, stream,
                       L.words, bitmap, prefix, dims, out_coors, out_capacity);
    OCOCC_CHECK_LAUNCH();
  }
  return OCOCC_OK;
}

namespace {
struct VoxLayout {
  Layout g;  // bitmap | prefix | scan scratch, exactly the grid_unique layout (so that
             // ococc_grid_unique_workspace_layout describes this workspace too)
  int64_t off_code, total;
};
inline bool make_vox_layout(int64_t n, int32_t batch, const int32_t* grid_zyx, VoxLayout* V) {
  if (n < 0 || batch < 1 || !grid_zyx) return false;
  const int32_t dims[4] = {batch, grid_zyx[0], grid_zyx[1], grid_zyx[2]};
  if (!make_layout(4, dims, &V->g)) return false;
  if ((V->g.words * 32) >= (1LL << 30)) return false;  // 2 * cell + dup must fit an int32
  V->off_code = V->g.total;
  V->total = V->off_code + ococc_align_up(n * 4, 256);
  return true;
}
}  // namespace

extern "C" int64_t ococc_voxelize_scatter_workspace_bytes(int64_t n, int32_t batch_size, const int32_t host_grid_zyx[3]) {
  VoxLayout V;
  if (!make_vox_layout(n, batch_size, host_grid_zyx, &V)) return -1;
  return V.total;
}

extern "C" int ococc_voxelize_scatter_mean_f32(const float* points, int32_t num_point_features,
                                               const int32_t* batch_idx, int64_t n, const float* feats, int32_t c,
                                               const float host_voxel_size[3], const float host_coors_range[6],
                                               int32_t batch_size, const int32_t host_grid_zyx[3],
                                               int32_t* voxel_coors, int64_t out_capacity, int32_t* inv,
                                               int32_t* counts, float* voxel_feats, uint16_t* voxel_feats_bf16,
                                               int32_t* num_voxels, int32_t* status, void* workspace,
                                               int64_t workspace_bytes, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VoxLayout V;
  OCOCC_REQUIRE(n >= 0 && out_capacity >= 0 && c >= 1, "bad sizes");
  OCOCC_REQUIRE(num_point_features >= 3, "num_point_features < 3");
  OCOCC_REQUIRE(host_voxel_size && host_coors_range, "null voxel_size / coors_range");
  OCOCC_REQUIRE(make_vox_layout(n, batch_size, host_grid_zyx, &V), "need batch >= 1, grid >= 1, batch*D*H*W < 2^30");
  OCOCC_REQUIRE(num_voxels && status && status == num_voxels + 1, "num_voxels/status: one device int32[2]");
  OCOCC_REQUIRE(voxel_feats, "voxel_feats (f32) is where the sums are kept: required");
  VoxGeom g;
  for (int i = 0; i < 3; ++i) {
    OCOCC_REQUIRE(host_voxel_size[i] > 0.f, "voxel size must be positive");
    // grid_size[i] = ceil((max - min) / voxel) in float, voxelization_cpu.cpp:155-158; (x,y,z) order
    const int gi = (int)ceilf((host_coors_range[3 + i] - host_coors_range[i]) / host_voxel_size[i]);
    OCOCC_REQUIRE(gi == host_grid_zyx[2 - i], "grid_zyx does not match coors_range / voxel_size");
  }
  g.vx = host_voxel_size[0]; g.vy = host_voxel_size[1]; g.vz = host_voxel_size[2];
  g.xmin = host_coors_range[0]; g.ymin = host_coors_range[1]; g.zmin = host_coors_range[2];
  g.gx = host_grid_zyx[2]; g.gy = host_grid_zyx[1]; g.gz = host_grid_zyx[0];
  g.batch = batch_size;
  if (out_capacity == 0 || n == 0) {  // nothing to mark: no voxels
    OCOCC_HIP(hipMemsetAsync(num_voxels, 0, 2 * sizeof(int32_t), stream));
    if (out_capacity == 0 && n == 0) return OCOCC_OK;
  }
  OCOCC_REQUIRE(workspace && workspace_bytes >= V.total, "workspace too small");
  OCOCC_REQUIRE(voxel_coors && counts && (n == 0 || (points && batch_idx && feats && inv)), "null device pointer");
  char* ws = (char*)workspace;
  uint32_t* bitmap = (uint32_t*)(ws + V.g.off_bitmap);
  uint32_t* prefix = (uint32_t*)(ws + V.g.off_prefix);
  uint32_t* scratch = (uint32_t*)(ws + V.g.off_scratch);
  int32_t* code_of = (int32_t*)(ws + V.off_code);
  // the only fill: status and num_voxels are (re)written by the launches themselves
  OCOCC_HIP(hipMemsetAsync(bitmap, 0, V.g.words * 4, stream));
  if (n > 0) {
    hipLaunchKernelGGL(voxel_mark_kernel, dim3(ococc_grid_1d(n, 256)), dim3(256), 0, stream, points,
                       (int)num_point_features, batch_idx, n, g, bitmap, code_of, status);
    OCOCC_CHECK_LAUNCH();
  }
  OCOCC_HIP(ococc_scan::exclusive_scan<ococc_scan::POPC>(bitmap, V.g.words, V.g.words, 1, prefix, V.g.words,
                                                         scratch, (uint32_t*)num_voxels, stream));
  const int pieces = (c + 3) / 4;
  const int64_t ba = ococc_cdiv(n * pieces, 256), bb = ococc_cdiv(V.g.words, 256),
                bc = ococc_cdiv(out_capacity * pieces, 256);
  OCOCC_REQUIRE(ba + bb + bc < 0x7fffffffLL, "too many workgroups");
  hipLaunchKernelGGL(voxel_emit_kernel, dim3((unsigned)(ba + bb + bc)), dim3(256), 0, stream, n, (int)c, pieces, feats,
                     bitmap, prefix, V.g.words, g, (int)ba, (int)bb, code_of, inv, voxel_coors, counts, voxel_feats,
                     voxel_feats_bf16, out_capacity, num_voxels, status);
  OCOCC_CHECK_LAUNCH();
  if (n > 0) {
    hipLaunchKernelGGL(voxel_dup_add_kernel, dim3(ococc_grid_1d(n, 256)), dim3(256), 0, stream, n, (int)c, feats, inv,
                       code_of, counts, voxel_feats, out_capacity);
    OCOCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(voxel_dup_mean_kernel, dim3(ococc_grid_1d(n * pieces, 256)), dim3(256), 0, stream, n, (int)c,
                       pieces, code_of, inv, counts, voxel_feats, voxel_feats_bf16);
    OCOCC_CHECK_LAUNCH();
  }
  return OCOCC_OK;
}
