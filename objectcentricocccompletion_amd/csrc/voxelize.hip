// B1 dynamic voxelisation for gfx950.
// Behaviour follows mmdet3d/ops/voxel/src/voxelization_cpu.cpp:8-41 (the CPU
// functor; the CUDA kernel voxelization_cuda.cu:25-65 computes the same):
// c = floor((p - min) / voxel) in float, clamped to [0, grid-1], stored (z,y,x).
// HBM-bound: 12 B read + 12 B written per point; one thread per point,
// grid-stride, no LDS needed.  Built WITHOUT fast-math so the float division
// and floor are IEEE and match the host bit for bit.
#include "common.hpp"

namespace {

__global__ void __launch_bounds__(256)
dynamic_voxelize_kernel(const float* __restrict__ points, int32_t* __restrict__ coors,
                        int64_t num_points, int num_features, float vx, float vy, float vz,
                        float xmin, float ymin, float zmin, int gx, int gy, int gz) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_points;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = points + i * num_features;
    int cx = (int)floorf((p[0] - xmin) / vx);
    int cy = (int)floorf((p[1] - ymin) / vy);
    int cz = (int)floorf((p[2] - zmin) / vz);
    cx = cx < 0 ? 0 : (cx >= gx ? gx - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= gy ? gy - 1 : cy);
    cz = cz < 0 ? 0 : (cz >= gz ? gz - 1 : cz);
    int32_t* c = coors + i * 3;
    c[0] = cz;
    c[1] = cy;
    c[2] = cx;
  }
}

}  // namespace

extern "C" int ococc_dynamic_voxelize_f32(const float* points, int64_t num_points,
                                          int32_t num_features, const float host_voxel_size[3],
                                          const float host_coors_range[6], int32_t* coors,
                                          ococc_stream_t stream) {
  OCOCC_REQUIRE(num_points >= 0, "num_points < 0");
  OCOCC_REQUIRE(num_features >= 3, "num_features < 3");
  OCOCC_REQUIRE(host_voxel_size && host_coors_range, "null voxel_size / coors_range");
  if (num_points == 0) return OCOCC_OK;
  OCOCC_REQUIRE(points && coors, "null device pointer");
  int g[3];
  for (int i = 0; i < 3; ++i) {
    OCOCC_REQUIRE(host_voxel_size[i] > 0.f, "voxel size must be positive");
    // grid_size[i] = ceil((max - min) / voxel) in float, voxelization_cpu.cpp:155-158
    g[i] = (int)ceilf((host_coors_range[3 + i] - host_coors_range[i]) / host_voxel_size[i]);
    OCOCC_REQUIRE(g[i] >= 1, "empty grid");
  }
  hipLaunchKernelGGL(dynamic_voxelize_kernel, dim3(ococc_grid_1d(num_points, 256)), dim3(256), 0,
                     (hipStream_t)stream, points, coors, num_points, (int)num_features,
                     host_voxel_size[0], host_voxel_size[1], host_voxel_size[2],
                     host_coors_range[0], host_coors_range[1], host_coors_range[2], g[0], g[1],
                     g[2]);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
