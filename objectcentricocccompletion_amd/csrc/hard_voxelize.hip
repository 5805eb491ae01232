// Hard voxelisation (max_points per voxel, max_voxels, first-come order).
// Reference: hard_voxelize_kernel, mmdet3d/ops/voxel/src/voxelization_cpu.cpp:44-99
// (sequential; the CUDA path voxelization_cuda.cu:110-184 is O(N^2) plus a
// <<<1,1>>> numbering kernel).  The sequential semantics are order dependent:
//   * a voxel's id is the rank of its FIRST point among all first points;
//   * voxels whose id >= max_voxels are dropped (later points of kept voxels
//     are still accepted);
//   * inside a voxel points keep index order and only the first max_points
//     are stored.
// Parallel restatement used here (deterministic, no sort):
//   bitmap + popcount scan -> compact rank r of every occupied cell;
//   first[r] = min point index (integer atomicMin), count[r];
//   voxel id = exclusive scan over points of [i == first[r(i)]];
//   slot(i)  = #{ j in members(r) : j < i }, members gathered through a
//              counting-sort cursor (list order is irrelevant to the count).
#include "common.hpp"
#include "scan.hpp"

namespace {

struct HV {
  float vx, vy, vz, xmin, ymin, zmin;
  int gx, gy, gz;
};

__device__ __forceinline__ int32_t hv_rank(const uint32_t* bitmap, const uint32_t* prefix,
                                           int64_t cell) {
  const uint32_t w = bitmap[cell >> 5];
  return (int32_t)(prefix[cell >> 5] + __popc(w & ((1u << (cell & 31)) - 1u)));
}

__global__ void __launch_bounds__(256)
hv_cells_kernel(const float* __restrict__ points, int64_t n, int nf, HV p,
                int32_t* __restrict__ cell_of, uint32_t* __restrict__ bitmap) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float* q = points + i * nf;
    int cx = (int)floorf((q[0] - p.xmin) / p.vx);
    int cy = (int)floorf((q[1] - p.ymin) / p.vy);
    int cz = (int)floorf((q[2] - p.zmin) / p.vz);
    cx = cx < 0 ? 0 : (cx >= p.gx ? p.gx - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= p.gy ? p.gy - 1 : cy);
    cz = cz < 0 ? 0 : (cz >= p.gz ? p.gz - 1 : cz);
    const int64_t cell = ((int64_t)cz * p.gy + cy) * p.gx + cx;
    cell_of[i] = (int32_t)cell;
    atomicOr(bitmap + (cell >> 5), 1u << (cell & 31));
  }
}

__global__ void __launch_bounds__(256)
hv_first_count_kernel(int64_t n, const int32_t* __restrict__ cell_of,
                      const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ prefix,
                      int32_t* __restrict__ rank_of, int32_t* __restrict__ first,
                      uint32_t* __restrict__ count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = hv_rank(bitmap, prefix, cell_of[i]);
    rank_of[i] = r;
    atomicMin(first + r, (int32_t)i);
    atomicAdd(count + r, 1u);
  }
}

__global__ void __launch_bounds__(256)
hv_members_kernel(int64_t n, const int32_t* __restrict__ rank_of, const int32_t* __restrict__ first,
                  const uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
                  int32_t* __restrict__ members, uint32_t* __restrict__ is_first) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = rank_of[i];
    members[offsets[r] + atomicAdd(cursor + r, 1u)] = (int32_t)i;
    is_first[i] = first[r] == (int32_t)i ? 1u : 0u;
  }
}

__global__ void __launch_bounds__(256)
hv_emit_kernel(const float* __restrict__ points, int64_t n, int nf, HV p,
               const int32_t* __restrict__ cell_of, const int32_t* __restrict__ rank_of,
               const int32_t* __restrict__ first, const uint32_t* __restrict__ count,
               const uint32_t* __restrict__ offsets, const int32_t* __restrict__ members,
               const uint32_t* __restrict__ vorder, int max_points, int max_voxels,
               float* __restrict__ voxels, int32_t* __restrict__ coors,
               int32_t* __restrict__ num_points_per_voxel) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = rank_of[i];
    const int32_t f = first[r];
    const uint32_t vid = vorder[f];
    if (max_voxels != -1 && (int64_t)vid >= max_voxels) continue;
    const uint32_t cnt = count[r];
    if (f == (int32_t)i) {
      int64_t cell = cell_of[i];
      const int cx = (int)(cell % p.gx);
      cell /= p.gx;
      const int cy = (int)(cell % p.gy);
      const int cz = (int)(cell / p.gy);
      coors[(int64_t)vid * 3 + 0] = cz;
      coors[(int64_t)vid * 3 + 1] = cy;
      coors[(int64_t)vid * 3 + 2] = cx;
      num_points_per_voxel[vid] =
          (max_points == -1 || cnt < (uint32_t)max_points) ? (int32_t)cnt : max_points;
    }
    uint32_t slot = 0;
    const int32_t* mem = members + offsets[r];
    for (uint32_t j = 0; j < cnt; ++j) slot += mem[j] < (int32_t)i ? 1u : 0u;
    if (max_points != -1 && slot >= (uint32_t)max_points) continue;
    float* dst = voxels + ((int64_t)vid * max_points + slot) * nf;
    const float* src = points + i * nf;
    for (int k = 0; k < nf; ++k) dst[k] = src[k];
  }
}

__global__ void hv_voxel_num_kernel(const uint32_t* total, int max_voxels, int32_t* voxel_num) {
  int32_t v = (int32_t)*total;
  if (max_voxels != -1 && v > max_voxels) v = max_voxels;
  *voxel_num = v;
}

struct Layout {
  int64_t words, o_bitmap, o_prefix, o_cell, o_rank, o_first, o_count, o_off, o_cursor, o_members,
      o_isfirst, o_vorder, o_scratch, o_total, total;
  int g[3];
};

inline bool make_layout(int64_t n, const float* vs, const float* range, Layout* L) {
  if (!vs || !range || n < 0) return false;
  int64_t cells = 1;
  for (int i = 0; i < 3; ++i) {
    if (!(vs[i] > 0.f)) return false;
    // grid_size[i] = round((max - min) / voxel), voxelization_cpu.cpp:119-122
    L->g[i] = (int)roundf((range[3 + i] - range[i]) / vs[i]);
    if (L->g[i] < 1) return false;
    cells *= L->g[i];
    if (cells > 0x7fffffffLL) return false;
  }
  L->words = (cells + 31) / 32;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { int64_t o = off; off += ococc_align_up(bytes, 256); return o; };
  L->o_bitmap = take(L->words * 4);
  L->o_prefix = take(L->words * 4);
  L->o_cell = take(n * 4);
  L->o_rank = take(n * 4);
  L->o_first = take(n * 4);
  L->o_count = take(n * 4);
  L->o_off = take(n * 4);
  L->o_cursor = take(n * 4);
  L->o_members = take(n * 4);
  L->o_isfirst = take(n * 4);
  L->o_vorder = take(n * 4);
  const int64_t s1 = ococc_scan::scratch_words(L->words, 1), s2 = ococc_scan::scratch_words(n, 1);
  L->o_scratch = take((s1 > s2 ? s1 : s2) * 4);
  L->o_total = take(16);
  L->total = off;
  return true;
}

}  // namespace

extern "C" int64_t ococc_hard_voxelize_workspace_bytes(int64_t num_points,
                                                       const float host_voxel_size[3],
                                                       const float host_coors_range[6]) {
  Layout L;
  if (!make_layout(num_points, host_voxel_size, host_coors_range, &L)) return -1;
  return L.total;
}

extern "C" int ococc_hard_voxelize_f32(const float* points, int64_t num_points,
                                       int32_t num_features, const float host_voxel_size[3],
                                       const float host_coors_range[6], int32_t max_points,
                                       int32_t max_voxels, float* voxels, int32_t* coors,
                                       int32_t* num_points_per_voxel, int32_t* voxel_num,
                                       void* workspace, int64_t workspace_bytes,
                                       ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Layout L;
  OCOCC_REQUIRE(make_layout(num_points, host_voxel_size, host_coors_range, &L),
                "bad voxel_size / coors_range (need positive sizes, grid < 2^31 cells)");
  OCOCC_REQUIRE(num_features >= 3, "num_features < 3");
  OCOCC_REQUIRE(max_points >= 1, "max_points must be >= 1 (use dynamic voxelisation otherwise)");
  OCOCC_REQUIRE(max_voxels == -1 || max_voxels >= 0, "max_voxels must be -1 or >= 0");
  OCOCC_REQUIRE(voxel_num, "null voxel_num");
  OCOCC_HIP(hipMemsetAsync(voxel_num, 0, sizeof(int32_t), stream));
  const int64_t n = num_points;
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(points && voxels && coors && num_points_per_voxel, "null pointer");
  OCOCC_REQUIRE(workspace && workspace_bytes >= L.total, "workspace too small");
  char* ws = (char*)workspace;
  uint32_t* bitmap = (uint32_t*)(ws + L.o_bitmap);
  uint32_t* prefix = (uint32_t*)(ws + L.o_prefix);
  int32_t* cell_of = (int32_t*)(ws + L.o_cell);
  int32_t* rank_of = (int32_t*)(ws + L.o_rank);
  int32_t* first = (int32_t*)(ws + L.o_first);
  uint32_t* count = (uint32_t*)(ws + L.o_count);
  uint32_t* offsets = (uint32_t*)(ws + L.o_off);
  uint32_t* cursor = (uint32_t*)(ws + L.o_cursor);
  int32_t* members = (int32_t*)(ws + L.o_members);
  uint32_t* is_first = (uint32_t*)(ws + L.o_isfirst);
  uint32_t* vorder = (uint32_t*)(ws + L.o_vorder);
  uint32_t* scratch = (uint32_t*)(ws + L.o_scratch);
  uint32_t* total = (uint32_t*)(ws + L.o_total);
  HV p{host_voxel_size[0], host_voxel_size[1], host_voxel_size[2], host_coors_range[0],
       host_coors_range[1], host_coors_range[2], L.g[0], L.g[1], L.g[2]};
  const int g1 = ococc_grid_1d(n, 256);
  OCOCC_HIP(hipMemsetAsync(bitmap, 0, L.words * 4, stream));
  OCOCC_HIP(hipMemsetAsync(first, 0x7f, n * 4, stream));
  OCOCC_HIP(hipMemsetAsync(count, 0, n * 4, stream));
  OCOCC_HIP(hipMemsetAsync(cursor, 0, n * 4, stream));
  hipLaunchKernelGGL(hv_cells_kernel, dim3(g1), dim3(256), 0, stream, points, n, (int)num_features,
                     p, cell_of, bitmap);
  OCOCC_CHECK_LAUNCH();
  OCOCC_HIP(ococc_scan::exclusive_scan<ococc_scan::POPC>(bitmap, L.words, L.words, 1, prefix,
                                                         L.words, scratch, total, stream));
  hipLaunchKernelGGL(hv_first_count_kernel, dim3(g1), dim3(256), 0, stream, n, cell_of, bitmap,
                     prefix, rank_of, first, count);
  OCOCC_CHECK_LAUNCH();
  // offsets over ranks (at most n occupied cells; the tail of count[] is zero)
  OCOCC_HIP(ococc_scan::exclusive_scan<ococc_scan::IDENT>(count, n, n, 1, offsets, n, scratch,
                                                          nullptr, stream));
  hipLaunchKernelGGL(hv_members_kernel, dim3(g1), dim3(256), 0, stream, n, rank_of, first, offsets,
                     cursor, members, is_first);
  OCOCC_CHECK_LAUNCH();
  OCOCC_HIP(ococc_scan::exclusive_scan<ococc_scan::IDENT>(is_first, n, n, 1, vorder, n, scratch,
                                                          nullptr, stream));
  hipLaunchKernelGGL(hv_emit_kernel, dim3(g1), dim3(256), 0, stream, points, n, (int)num_features,
                     p, cell_of, rank_of, first, count, offsets, members, vorder, (int)max_points,
                     (int)max_voxels, voxels, coors, num_points_per_voxel);
  OCOCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(hv_voxel_num_kernel, dim3(1), dim3(1), 0, stream, total, (int)max_voxels,
                     voxel_num);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
