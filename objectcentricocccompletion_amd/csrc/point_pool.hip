// A3 points-in-rotated-box pooling (the TorchEx dynamic_point_pool_mixed replacement).
// Reference call site: mmdet3d/ops/dynamic_point_pool_op.py:63-113 and
// models/roi_heads/roi_extractors/dynamic_point_roi_extractor.py:177-243; the CUDA source
// (Abyssaledge/TorchEx) is not vendored, so the contract is taken from the call site and
// its debug assertions (:222-234):
//   * a (point, RoI) pair is emitted when their int keys are equal (key = batch*max_frames
//     + frame) and the point lies in the RoI enlarged by extra_wlh;
//   * 13 floats per pair: xyz, box-frame xyz (x along l, y along w, origin at the box
//     centre, WITHOUT the pi/2 fix the callers apply later), the six distances to the faces
//     of the ORIGINAL box (off[0]+off[3]=l, off[1]+off[4]=w, off[2]+off[5]=h), is_in_margin;
//   * at most max_inbox_point pairs per RoI and max_all_pts pairs in total.
// Our definitions where the contract is silent (documented, "parity unpinned"):
//   enlarged size = size + extra (extra/2 per side); box test strict in x/y, inclusive in
//   z (as mmdet3d's check_pt_in_box3d); is_in_margin = inside the enlarged box but not the
//   original one; when a cap bites the pairs with the SMALLEST point indices are kept.
//
// MI355X design: the reference kernel hands out output slots with atomicAdd, so its row
// order is "sorted by RoI, not strictly guaranteed".  Here rows come out exactly sorted by
// (RoI, point index): a count pass, two tiny prefix passes and a write pass that ranks
// pairs inside a 1024-point chunk with wave ballots + popcounts.  Sorted rows make every
// later segment reduction a run-length reduction (scatter_reduce.hip) and the result is
// deterministic.  Work: N x R cheap tests done twice; RoI tiles of 16 are staged in LDS so
// each point is loaded once per tile.
#include "common.hpp"
#include "scan.hpp"

namespace {

constexpr int kPtsPerBlock = 1024;  // 256 threads x 4 points (point = base + j*256 + tid)
constexpr int kRoiTile = 16;

struct RoiLDS {
  float cx, cy, cz, hw, hl, hh, cosa, sina, ehw, ehl, ehh;
  int key;
};

__device__ __forceinline__ void load_roi(const float* __restrict__ rois, const int32_t* __restrict__ keys,
                                         int r, int R, float ew, float el, float eh, RoiLDS* d) {
  if (r < R) {
    const float* b = rois + (int64_t)r * 7;
    const float w = b[3], l = b[4], h = b[5], rz = b[6];
    d->cx = b[0];
    d->cy = b[1];
    d->cz = b[2] + h * 0.5f;  // bottom centre -> gravity centre
    d->hw = w * 0.5f;
    d->hl = l * 0.5f;
    d->hh = h * 0.5f;
    d->ehw = (w + ew) * 0.5f;
    d->ehl = (l + el) * 0.5f;
    d->ehh = (h + eh) * 0.5f;
    d->cosa = cosf(-rz);
    d->sina = sinf(-rz);
    d->key = keys[r];
  } else {
    d->key = -0x7fffffff;
    d->ehw = d->ehl = d->ehh = -1.f;
  }
}

// 0 = outside, 1 = inside the original box, 2 = only inside the enlarged box
__device__ __forceinline__ int test_point(const RoiLDS& b, float x, float y, float z, float* lx,
                                          float* ly, float* lz) {
  const float dz = z - b.cz;
  if (fabsf(dz) > b.ehh) return 0;
  const float sx = x - b.cx, sy = y - b.cy;
  const float px = sx * b.cosa + sy * (-b.sina);
  const float py = sx * b.sina + sy * b.cosa;
  if (!(px > -b.ehl && px < b.ehl && py > -b.ehw && py < b.ehw)) return 0;
  *lx = px;
  *ly = py;
  *lz = dz;
  const bool inner = fabsf(dz) <= b.hh && px > -b.hl && px < b.hl && py > -b.hw && py < b.hw;
  return inner ? 1 : 2;
}

// WRITE = false: counts[r][chunk].  WRITE = true: emit rows.
template <bool WRITE>
__global__ void __launch_bounds__(256)
pool_kernel(const float* __restrict__ rois, const int32_t* __restrict__ roi_key, int R,
            const float* __restrict__ pts, const int32_t* __restrict__ pts_key, int64_t N, float ew,
            float el, float eh, int chunks, uint32_t* __restrict__ counts,
            const uint32_t* __restrict__ roi_prefix,  // [R][chunks] exclusive, uncapped
            const uint32_t* __restrict__ roi_base,    // [R] exclusive over capped totals
            int max_inbox, int64_t max_all, int64_t* __restrict__ out_pts_idx,
            int64_t* __restrict__ out_roi_idx, float* __restrict__ out_feats) {
  __shared__ RoiLDS tile[kRoiTile];
  __shared__ uint32_t wcnt[kRoiTile][4][4];  // [roi][j][wave]
  const int chunk = blockIdx.x;
  const int r0 = blockIdx.y * kRoiTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < kRoiTile) load_roi(rois, roi_key, r0 + threadIdx.x, R, ew, el, eh, &tile[threadIdx.x]);
  __syncthreads();
  float px[4], py[4], pz[4];
  int pk[4];
  int64_t pi[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    pi[j] = (int64_t)chunk * kPtsPerBlock + j * 256 + threadIdx.x;
    if (pi[j] < N) {
      px[j] = pts[pi[j] * 3];
      py[j] = pts[pi[j] * 3 + 1];
      pz[j] = pts[pi[j] * 3 + 2];
      pk[j] = pts_key[pi[j]];
    } else {
      px[j] = py[j] = pz[j] = 0.f;
      pk[j] = 0x7fffffff;  // matches no RoI key
    }
  }
  for (int q = 0; q < kRoiTile; ++q) {
    const RoiLDS b = tile[q];
    int flag[4];
    float lx[4], ly[4], lz[4];
    unsigned long long bal[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      flag[j] = 0;
      if (pk[j] == b.key) flag[j] = test_point(b, px[j], py[j], pz[j], &lx[j], &ly[j], &lz[j]);
      bal[j] = __ballot(flag[j] != 0);
      if (lane == 0) wcnt[q][j][wave] = (uint32_t)__popcll(bal[j]);
    }
    if (!WRITE) continue;
    __syncthreads();  // wcnt[q] complete (uniform: every thread runs every q)
    const int r = r0 + q;
    if (r >= R) continue;
    const uint32_t before_chunk = roi_prefix[(int64_t)r * chunks + chunk];
    const uint32_t base = roi_base[r];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!flag[j]) continue;
      uint32_t rank = before_chunk;
      for (int jj = 0; jj < 4; ++jj)
        for (int w = 0; w < 4; ++w)
          if (jj < j || (jj == j && w < wave)) rank += wcnt[q][jj][w];
      rank += (uint32_t)__popcll(bal[j] & ((1ull << lane) - 1ull));
      if ((int)rank >= max_inbox) continue;   // per-RoI cap: keep the smallest point indices
      const int64_t pos = (int64_t)base + rank;
      if (pos >= max_all) continue;           // global cap
      out_pts_idx[pos] = pi[j];
      out_roi_idx[pos] = r;
      float* f = out_feats + pos * 13;
      f[0] = px[j]; f[1] = py[j]; f[2] = pz[j];
      f[3] = lx[j]; f[4] = ly[j]; f[5] = lz[j];
      f[6] = lx[j] + b.hl; f[7] = ly[j] + b.hw; f[8] = lz[j] + b.hh;
      f[9] = b.hl - lx[j]; f[10] = b.hw - ly[j]; f[11] = b.hh - lz[j];
      f[12] = flag[j] == 2 ? 1.f : 0.f;
    }
  }
  if (!WRITE) {
    __syncthreads();
    if (threadIdx.x < kRoiTile && r0 + threadIdx.x < R) {
      uint32_t s = 0;
      for (int j = 0; j < 4; ++j)
        for (int w = 0; w < 4; ++w) s += wcnt[threadIdx.x][j][w];
      counts[(int64_t)(r0 + threadIdx.x) * chunks + chunk] = s;
    }
  }
}

// one thread per RoI: exclusive prefix over its chunks (uncapped) and capped total
__global__ void __launch_bounds__(256)
roi_scan_kernel(const uint32_t* __restrict__ counts, int R, int chunks, int max_inbox,
                uint32_t* __restrict__ roi_prefix, uint32_t* __restrict__ capped_total) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  uint32_t run = 0;
  for (int c = 0; c < chunks; ++c) {
    roi_prefix[(int64_t)r * chunks + c] = run;
    run += counts[(int64_t)r * chunks + c];
  }
  capped_total[r] = run < (uint32_t)max_inbox ? run : (uint32_t)max_inbox;
}

__global__ void finish_kernel(const uint32_t* __restrict__ total, int64_t max_all,
                              const uint32_t* __restrict__ capped_total,
                              const uint32_t* __restrict__ roi_base, int R,
                              int32_t* __restrict__ roi_counts, int32_t* __restrict__ num_out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0) *num_out = (int32_t)((int64_t)*total < max_all ? (int64_t)*total : max_all);
  if (r < R && roi_counts) {
    int64_t lo = roi_base[r], hi = lo + capped_total[r];
    if (lo > max_all) lo = max_all;
    if (hi > max_all) hi = max_all;
    roi_counts[r] = (int32_t)(hi - lo);
  }
}

struct Layout {
  int chunks;
  int64_t o_counts, o_prefix, o_capped, o_base, o_scratch, o_total, total;
};
inline bool make_layout(int64_t n, int64_t r, Layout* L) {
  if (n < 0 || r < 0 || n > 0x7fffffffLL || r > 0x7fffffffLL) return false;
  L->chunks = (int)ococc_cdiv(n > 0 ? n : 1, kPtsPerBlock);
  int64_t off = 0;
  auto take = [&](int64_t b) { int64_t o = off; off += ococc_align_up(b > 0 ? b : 4, 256); return o; };
  L->o_counts = take(r * L->chunks * 4);
  L->o_prefix = take(r * L->chunks * 4);
  L->o_capped = take(r * 4);
  L->o_base = take(r * 4);
  L->o_scratch = take(ococc_scan::scratch_words(r > 0 ? r : 1, 1) * 4);
  L->o_total = take(16);
  L->total = off;
  return true;
}

}  // namespace

extern "C" int64_t ococc_point_pool_workspace_bytes(int64_t num_points, int64_t num_rois) {
  Layout L;
  if (!make_layout(num_points, num_rois, &L)) return -1;
  return L.total;
}

extern "C" int ococc_dynamic_point_pool_mixed(const float* rois, const int32_t* rois_key,
                                              int64_t num_rois, const float* pts,
                                              const int32_t* pts_key, int64_t num_points,
                                              const float host_extra_wlh[3], int32_t max_inbox_point,
                                              int64_t max_all_pts, int64_t* out_pts_idx,
                                              int64_t* out_roi_idx, float* out_pts_feats,
                                              int32_t* roi_counts, int32_t* num_out,
                                              void* workspace, int64_t workspace_bytes,
                                              ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Layout L;
  OCOCC_REQUIRE(make_layout(num_points, num_rois, &L), "bad sizes");
  OCOCC_REQUIRE(host_extra_wlh, "null extra_wlh");
  OCOCC_REQUIRE(max_inbox_point >= 1 && max_all_pts >= 0, "caps must be positive");
  OCOCC_REQUIRE(num_out, "null num_out");
  OCOCC_HIP(hipMemsetAsync(num_out, 0, sizeof(int32_t), stream));
  if (roi_counts && num_rois > 0) OCOCC_HIP(hipMemsetAsync(roi_counts, 0, num_rois * 4, stream));
  if (num_rois == 0 || num_points == 0 || max_all_pts == 0) return OCOCC_OK;
  OCOCC_REQUIRE(rois && rois_key && pts && pts_key, "null input");
  OCOCC_REQUIRE(out_pts_idx && out_roi_idx && out_pts_feats, "null output");
  OCOCC_REQUIRE(workspace && workspace_bytes >= L.total, "workspace too small");
  char* ws = (char*)workspace;
  uint32_t* counts = (uint32_t*)(ws + L.o_counts);
  uint32_t* prefix = (uint32_t*)(ws + L.o_prefix);
  uint32_t* capped = (uint32_t*)(ws + L.o_capped);
  uint32_t* base = (uint32_t*)(ws + L.o_base);
  uint32_t* scratch = (uint32_t*)(ws + L.o_scratch);
  uint32_t* total = (uint32_t*)(ws + L.o_total);
  const int R = (int)num_rois;
  // extra_wlh is ordered (w, l, h) like the box sizes
  const float ew = host_extra_wlh[0], el = host_extra_wlh[1], eh = host_extra_wlh[2];
  dim3 grid(L.chunks, (unsigned)ococc_cdiv(R, kRoiTile));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_kernel<false>), grid, dim3(256), 0, stream, rois, rois_key,
                     R, pts, pts_key, num_points, ew, el, eh, L.chunks, counts,
                     (const uint32_t*)nullptr, (const uint32_t*)nullptr, (int)max_inbox_point,
                     max_all_pts, (int64_t*)nullptr, (int64_t*)nullptr, (float*)nullptr);
  OCOCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(roi_scan_kernel, dim3((unsigned)ococc_cdiv(R, 256)), dim3(256), 0, stream, counts,
                     R, L.chunks, (int)max_inbox_point, prefix, capped);
  OCOCC_CHECK_LAUNCH();
  OCOCC_HIP(ococc_scan::exclusive_scan<ococc_scan::IDENT>(capped, R, R, 1, base, R, scratch, total,
                                                          stream));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_kernel<true>), grid, dim3(256), 0, stream, rois, rois_key, R,
                     pts, pts_key, num_points, ew, el, eh, L.chunks, counts, prefix, base,
                     (int)max_inbox_point, max_all_pts, out_pts_idx, out_roi_idx, out_pts_feats);
  OCOCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(finish_kernel, dim3((unsigned)ococc_cdiv(R, 256)), dim3(256), 0, stream, total,
                     max_all_pts, capped, base, R, roi_counts, num_out);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
