// A12 glue: the small element-wise chains around the targets and the occupancy samples as single launches.  At the
// config's own batch (4 tracklets, 128 RoIs) each of these chains was a dozen to three dozen ATen launches of a few
// hundred elements (tools/aten_sites_b4.py): the step is bound by the host there.
//   ococc_rotate_z_f32            rotation_3d_in_axis(points, angles, axis=2)   mmdet3d/core/bbox/structures/utils.py:21-61
//                                 (the reference multiplies by the transposed matrix: einsum 'aij,jka->aik')
//   ococc_points_box_to_box_f32   points of box A's frame (gravity centred) into box B's frame:
//                                 rotate by yaw_A, + centre_A, z + h_A / 2, - centre_B, z - h_B / 2, rotate by -yaw_B
//                                 mmdet3d/models/roi_heads/bbox_heads/ococc_bbox_head.py:1279-1290 (targets), 714-724 (loss_occ),
//                                 mmdet3d/models/roi_heads/tracklet_roi_head_occ.py (test_occ)
//   ococc_roi_box_targets_f32     GT boxes in the canonical frame of their RoIs -> DeltaXYZWLHRBBoxCoder deltas
//                                 ococc_bbox_head.py:1190-1222 + mmdet3d/core/bbox/coders/delta_xyzwhlr_bbox_coder.py:21-50
// f32, the operations and their order as the torch expressions they replace (sums of products in the order of the matrix
// product's k index; remainders with the sign of the divisor).
#include "common.hpp"

namespace {

__device__ __forceinline__ float remainder_pos(float x, float m) {   // torch.remainder for m > 0
  float r = fmodf(x, m);
  if (r != 0.f && r < 0.f) r += m;
  return r;
}

// out = p R^T-convention of the reference for axis 2:  x' = x c + y s,  y' = -x s + y c,  z' = z
__device__ __forceinline__ void rot_z(float x, float y, float z, float s, float c, float& ox, float& oy, float& oz) {
  ox = fmaf(z, 0.f, fmaf(y, s, x * c));
  oy = fmaf(z, 0.f, fmaf(y, c, x * -s));
  oz = fmaf(z, 1.f, fmaf(y, 0.f, x * 0.f));
}

__global__ void __launch_bounds__(256)
rotate_z_kernel(const float* __restrict__ p, const float* __restrict__ ang, int64_t n, int64_t m, float* __restrict__ out) {
  const int64_t total = n * m;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t a = i / m;
    float s, c;
    sincosf(ang[a], &s, &c);
    rot_z(p[3 * i], p[3 * i + 1], p[3 * i + 2], s, c, out[3 * i], out[3 * i + 1], out[3 * i + 2]);
  }
}

__global__ void __launch_bounds__(256)
points_box_to_box_kernel(const float* __restrict__ p, const float* __restrict__ from, int64_t ld_from,
                         const float* __restrict__ to, int64_t ld_to, int64_t n, int64_t m, float* __restrict__ out) {
  const int64_t total = n * m;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t a = i / m;
    const float* fb = from + a * ld_from;
    const float* tb = to + a * ld_to;
    float s, c, x, y, z;
    sincosf(fb[6], &s, &c);
    rot_z(p[3 * i], p[3 * i + 1], p[3 * i + 2], s, c, x, y, z);
    x += fb[0];
    y += fb[1];
    z += fb[2];
    z += fb[5] / 2.f;
    x -= tb[0];
    y -= tb[1];
    z -= tb[2];
    z -= tb[5] / 2.f;
    sincosf(-tb[6], &s, &c);
    rot_z(x, y, z, s, c, out[3 * i], out[3 * i + 1], out[3 * i + 2]);
  }
}

__global__ void __launch_bounds__(256)
roi_box_targets_kernel(const float* __restrict__ roi, int64_t ld_roi, const float* __restrict__ gt, int64_t ld_gt, int64_t n,
                       float* __restrict__ out) {
  constexpr float kPi = 3.14159265358979323846f, kTwoPi = 6.28318530717958647692f, kHalfPi = 1.57079632679489661923f;
  constexpr float kThreeHalfPi = 4.71238898038468985769f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float* r = roi + i * ld_roi;
    const float* g = gt + i * ld_gt;
    const float roi_ry = remainder_pos(r[6], kTwoPi);
    float x = g[0] - r[0], y = g[1] - r[1], z = g[2] - r[2];
    float ry = g[6] - roi_ry;
    float s, c, cx, cy, cz;
    sincosf(-(roi_ry + kHalfPi), &s, &c);
    rot_z(x, y, z, s, c, cx, cy, cz);
    ry = remainder_pos(ry, kTwoPi);
    if (ry > kHalfPi && ry < kThreeHalfPi) ry = remainder_pos(ry + kPi, kTwoPi);
    if (ry > kPi) ry -= kTwoPi;
    ry = fminf(fmaxf(ry, -kHalfPi), kHalfPi);
    // DeltaXYZWLHRBBoxCoder.encode(anchor = the RoI at the origin with yaw 0, gt in its frame)
    const float wa = r[3], la = r[4], ha = r[5], wg = g[3], lg = g[4], hg = g[5];
    const float za = 0.f + ha / 2.f, zg = cz + hg / 2.f;
    const float diagonal = sqrtf(la * la + wa * wa);
    float* o = out + i * 7;
    o[0] = (cx - 0.f) / diagonal;
    o[1] = (cy - 0.f) / diagonal;
    o[2] = (zg - za) / ha;
    o[3] = logf(wg / wa);
    o[4] = logf(lg / la);
    o[5] = logf(hg / ha);
    o[6] = ry - 0.f;
  }
}

}  // namespace

extern "C" int ococc_rotate_z_f32(const float* points, const float* angles, int64_t n, int64_t m, float* out,
                                  ococc_stream_t stream) {
  OCOCC_REQUIRE(n >= 0 && m >= 0, "negative sizes");
  if (n * m == 0) return OCOCC_OK;
  OCOCC_REQUIRE(points && angles && out, "null pointer");
  hipLaunchKernelGGL(rotate_z_kernel, dim3(ococc_grid_1d(n * m, 256, 2048)), dim3(256), 0, (hipStream_t)stream, points, angles, n, m,
                     out);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_points_box_to_box_f32(const float* points, const float* from_boxes, int64_t ld_from,
                                           const float* to_boxes, int64_t ld_to, int64_t n, int64_t m, float* out,
                                           ococc_stream_t stream) {
  OCOCC_REQUIRE(n >= 0 && m >= 0 && ld_from >= 7 && ld_to >= 7, "bad sizes (boxes are rows of >= 7 floats)");
  if (n * m == 0) return OCOCC_OK;
  OCOCC_REQUIRE(points && from_boxes && to_boxes && out, "null pointer");
  hipLaunchKernelGGL(points_box_to_box_kernel, dim3(ococc_grid_1d(n * m, 256, 2048)), dim3(256), 0, (hipStream_t)stream, points,
                     from_boxes, ld_from, to_boxes, ld_to, n, m, out);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_roi_box_targets_f32(const float* rois, int64_t ld_rois, const float* gt_boxes, int64_t ld_gt, int64_t n,
                                         float* out, ococc_stream_t stream) {
  OCOCC_REQUIRE(n >= 0 && ld_rois >= 7 && ld_gt >= 7, "bad sizes (boxes are rows of >= 7 floats)");
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(rois && gt_boxes && out, "null pointer");
  hipLaunchKernelGGL(roi_box_targets_kernel, dim3(ococc_grid_1d(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, rois, ld_rois,
                     gt_boxes, ld_gt, n, out);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
