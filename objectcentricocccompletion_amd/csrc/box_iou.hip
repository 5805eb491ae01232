// A2 pairwise ("1 to 1") 3-D IoU of rotated boxes: the arithmetic of
// LiDARInstance3DBoxes.aligned_iou_3d (mmdet3d/core/bbox/structures/lidar_box3d.py:404-448),
// whose BEV part is TorchEx boxes_overlap_1to1 (source not vendored; we follow the iou3d
// convention the fork's own iou3d op uses: BEV rectangle (x,y,w,l) with w along x at yaw 0,
// corners turned CLOCKWISE by yaw).  IoU = inter_bev * overlap_h / max(v1 + v2 - inter, 1e-8).
// One thread per pair: Sutherland-Hodgman clipping of two convex quads (<= 8 vertices) and the
// shoelace area.  A few hundred pairs per step: latency only, no roofline to speak of.
#include "common.hpp"

namespace {

struct P2 { float x, y; };

__device__ __forceinline__ void corners(const float* b, P2* c) {
  const float cx = b[0], cy = b[1], hw = b[3] * 0.5f, hl = b[4] * 0.5f;
  const float ca = cosf(b[6]), sa = sinf(b[6]);
  const float dx[4] = {-hw, hw, hw, -hw}, dy[4] = {-hl, -hl, hl, hl};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c[k].x = dx[k] * ca + dy[k] * sa + cx;
    c[k].y = -dx[k] * sa + dy[k] * ca + cy;
  }
}

__device__ __forceinline__ float cross3(P2 a, P2 b, P2 p) {
  return (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x);
}

__device__ float quad_intersection_area(const P2* A, const P2* B) {
  P2 poly[16], tmp[16];
  int n = 4;
  for (int i = 0; i < 4; ++i) poly[i] = A[i];
  // orientation of B (corners() gives a consistent but convention dependent winding)
  float orient = 0.f;
  for (int i = 0; i < 4; ++i) orient += B[i].x * B[(i + 1) & 3].y - B[(i + 1) & 3].x * B[i].y;
  const float sgn = orient >= 0.f ? 1.f : -1.f;
  for (int e = 0; e < 4 && n > 0; ++e) {
    const P2 a = B[e], b = B[(e + 1) & 3];
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const P2 p = poly[i], q = poly[(i + 1) % n];
      const float dp = sgn * cross3(a, b, p), dq = sgn * cross3(a, b, q);
      if (dp >= 0.f) tmp[m++] = p;
      if ((dp >= 0.f) != (dq >= 0.f)) {
        const float t = dp / (dp - dq);
        tmp[m].x = p.x + t * (q.x - p.x);
        tmp[m].y = p.y + t * (q.y - p.y);
        ++m;
      }
    }
    n = m;
    for (int i = 0; i < n; ++i) poly[i] = tmp[i];
  }
  float area = 0.f;
  for (int i = 0; i < n; ++i) area += poly[i].x * poly[(i + 1) % n].y - poly[(i + 1) % n].x * poly[i].y;
  return fabsf(area) * 0.5f;
}

__global__ void __launch_bounds__(256)
aligned_iou3d_kernel(const float* __restrict__ b1, const float* __restrict__ b2, int64_t n,
                     float* __restrict__ iou) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* a = b1 + i * 7;
  const float* b = b2 + i * 7;
  P2 ca[4], cb[4];
  corners(a, ca);
  corners(b, cb);
  const float inter_bev = quad_intersection_area(ca, cb);
  const float top = fminf(a[2] + a[5], b[2] + b[5]), bot = fmaxf(a[2], b[2]);
  const float oh = fmaxf(top - bot, 0.f);
  const float inter = inter_bev * oh;
  const float v1 = a[3] * a[4] * a[5], v2 = b[3] * b[4] * b[5];
  iou[i] = inter / fmaxf(v1 + v2 - inter, 1e-8f);
}

}  // namespace

extern "C" int ococc_aligned_iou3d_f32(const float* boxes1, const float* boxes2, int64_t n,
                                       float* iou, ococc_stream_t stream) {
  OCOCC_REQUIRE(n >= 0, "n < 0");
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(boxes1 && boxes2 && iou, "null pointer");
  hipLaunchKernelGGL(aligned_iou3d_kernel, dim3((unsigned)ococc_cdiv(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, boxes1, boxes2, n, iou);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
