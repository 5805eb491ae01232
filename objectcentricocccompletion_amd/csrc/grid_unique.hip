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
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t* c = coors + i * dims.ndim;
    int64_t cell = 0;
    bool neg = false, over = false;
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
    if (neg) {
      cell_of[i] = -1;
    } else {
      cell_of[i] = (int32_t)cell;
      atomicOr(bitmap + (cell >> 5), 1u << (cell & 31));
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
