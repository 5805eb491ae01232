// A5 / B2 segment (scatter) reduce: max / mean / sum of point rows into voxel or
// RoI rows, forward and backward.
// Reference behaviour: feats_reduce_kernel and the traceback kernels of
// mmdet3d/ops/voxel/src/scatter_points_cuda.cu:81-179 (thread per point,
// per-channel CAS-max / atomicAdd) and torch_scatter.scatter_max / scatter
// called from scatter_v2 (mmdet3d/ops/sst/sst_ops.py:171-174).
//
// MI355X design: HBM-bound (n*c*4 B read, the output is tiny).  A wave reads
// whole rows (lane = channel, 256 B per wave instruction), keeps a running
// value while consecutive rows stay in the same segment and issues one atomic
// per (segment run, channel): rows arrive grouped by RoI in the OcOccNet path,
// so almost all of the reduction happens in registers and the float atomics
// that remain are full 256 B wave instructions (guide: Guideline 12).
// MAX uses integer atomics on the float bit pattern: exact and order
// independent.  SUM/MEAN use global_atomic_add_f32: order dependent in the
// last bits, like the reference's own GPU path.
#include "common.hpp"

namespace {

constexpr int kMaxRowsPerWave = 64;

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  // sign-split trick: valid because the destination starts at -inf
  unsigned int bits = __float_as_uint(v);
  if (!(bits >> 31))
    atomicMax((int*)addr, (int)bits);
  else
    atomicMin((unsigned int*)addr, bits);
}

// counts (optional): a segment with exactly one member needs no atomic -- a plain store of its only value.
// In voxelised point clouds almost every voxel holds one point (98.5 % in the 0.2 m benchmark grids), and
// plain stores run an order of magnitude faster than float atomics.
template <int MODE>
__device__ __forceinline__ void flush(float* out, int64_t seg, int c, int ch, float acc,
                                      const int32_t* __restrict__ counts) {
  float* dst = out + seg * c + ch;
  if (counts && counts[seg] == 1) {
    *dst = acc;
    return;
  }
  if (MODE == OCOCC_REDUCE_MAX)
    atomic_max_f32(dst, acc);
  else
    atomicAdd(dst, acc);
}

// cp = lanes per row (power of two <= 64); sub = 64 / cp rows read per step.
template <int MODE>
__global__ void __launch_bounds__(256)
segment_reduce_kernel(const float* __restrict__ feats, const int32_t* __restrict__ inv, int64_t n,
                      int c, int cp, int kRowsPerWave, float* __restrict__ out,
                      const int32_t* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int sub = 64 / cp;
  const int s = lane / cp, ch0 = lane % cp;
  for (int64_t r0 = wave * kRowsPerWave; r0 < n; r0 += nwaves * kRowsPerWave) {
    const int64_t r1 = (r0 + kRowsPerWave < n) ? r0 + kRowsPerWave : n;
    for (int ch = ch0; ch < c; ch += cp) {
      int32_t cur = -1;
      float acc = 0.f;
      // four rows per trip, their loads issued together: the walk is a chain of dependent compares, and with one row
      // per trip every step paid a global round trip (96 us for 131 k rows x 128 channels, 0.7 TB/s)
      for (int64_t r = r0 + s; r < r1; r += 4 * sub) {
        int32_t sg[4];
        float vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t rr = r + (int64_t)u * sub;
          const bool ok = rr < r1;
          sg[u] = ok ? inv[rr] : -1;
          vv[u] = ok ? feats[rr * c + ch] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int32_t seg = sg[u];
          if (seg < 0) continue;
          const float v = vv[u];
          if (seg != cur) {
            if (cur >= 0) flush<MODE>(out, cur, c, ch, acc, counts);
            cur = seg;
            acc = v;
          } else {
            acc = (MODE == OCOCC_REDUCE_MAX) ? fmaxf(acc, v) : acc + v;
          }
        }
      }
      if (cur >= 0) flush<MODE>(out, cur, c, ch, acc, counts);
    }
  }
}

// Short segments (voxelisation: ~1 point per voxel, rows in random order): the run-length pass above is a
// serial walk per lane group that never merges anything; one thread per element is fully parallel.
template <int MODE>
__global__ void __launch_bounds__(256)
segment_scatter_elem_kernel(const float* __restrict__ feats, const int32_t* __restrict__ inv, int64_t n, int c,
                            float* __restrict__ out, const int32_t* __restrict__ counts) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * c; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c;
    const int32_t seg = inv[r];
    if (seg < 0) continue;
    flush<MODE>(out, seg, c, (int)(i - r * c), feats[i], counts);
  }
}

// out = v (and arg = 0x7f7f7f7f when given) in one launch
__global__ void __launch_bounds__(256) fill_f32_kernel(float* p, int32_t* arg, int64_t count, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    p[i] = v;
    if (arg) arg[i] = 0x7f7f7f7f;
  }
}

// MEAN: divide by the count; MAX: segments nobody wrote become 0 (torch_scatter)
template <int MODE>
__global__ void __launch_bounds__(256)
finalize_kernel(float* __restrict__ out, const int32_t* __restrict__ counts, int64_t segs, int c) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < segs * c;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t cnt = counts[i / c];
    if (MODE == OCOCC_REDUCE_MEAN)
      out[i] = cnt > 0 ? out[i] / (float)cnt : 0.f;
    else if (cnt == 0)
      out[i] = 0.f;
  }
}

__global__ void __launch_bounds__(256)
argmax_kernel(const float* __restrict__ feats, const int32_t* __restrict__ inv, int64_t n, int c,
              const float* __restrict__ out, int32_t* __restrict__ arg) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * c;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c;
    const int ch = (int)(i - r * c);
    const int32_t seg = inv[r];
    if (seg < 0) continue;
    if (feats[i] == out[(int64_t)seg * c + ch]) atomicMin(arg + (int64_t)seg * c + ch, (int32_t)r);
  }
}

__global__ void __launch_bounds__(256)
segment_count_kernel(const int32_t* __restrict__ inv, int64_t n, int32_t* __restrict__ counts) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t seg = inv[i];
    if (seg >= 0) atomicAdd(counts + seg, 1);
  }
}

template <int MODE>
__global__ void __launch_bounds__(256)
segment_reduce_bwd_kernel(const float* __restrict__ go, const int32_t* __restrict__ inv, int64_t n,
                          int c, const int32_t* __restrict__ counts,
                          const int32_t* __restrict__ arg, float* __restrict__ gi) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * c;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c;
    const int ch = (int)(i - r * c);
    const int32_t seg = inv[r];
    float g = 0.f;
    if (seg >= 0) {
      const int64_t o = (int64_t)seg * c + ch;
      if (MODE == OCOCC_REDUCE_SUM)
        g = go[o];
      else if (MODE == OCOCC_REDUCE_MEAN)
        g = go[o] / (float)counts[seg];
      else
        g = (arg[o] == (int32_t)r) ? go[o] : 0.f;
    }
    gi[i] = g;
  }
}

inline int lanes_per_row(int c) {
  int cp = 1;
  while (cp < c && cp < 64) cp <<= 1;
  return cp;
}

}  // namespace

extern "C" int ococc_segment_count_i32(const int32_t* inv, int64_t n, int32_t* counts,
                                       int64_t num_segments, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(n >= 0 && num_segments >= 0, "negative size");
  if (num_segments > 0) {
    OCOCC_REQUIRE(counts, "null counts");
    OCOCC_HIP(hipMemsetAsync(counts, 0, num_segments * sizeof(int32_t), stream));
  }
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(inv, "null inv");
  hipLaunchKernelGGL(segment_count_kernel, dim3(ococc_grid_1d(n, 256)), dim3(256), 0, stream, inv,
                     n, counts);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_segment_reduce_f32(const float* feats, const int32_t* inv, int64_t n,
                                        int32_t c, int32_t reduce_type, const int32_t* counts,
                                        float* out, int32_t* arg, int64_t num_segments,
                                        ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(n >= 0 && num_segments >= 0 && c >= 1, "bad sizes");
  OCOCC_REQUIRE(reduce_type >= 0 && reduce_type <= 2, "reduce_type must be SUM/MEAN/MAX");
  OCOCC_REQUIRE(reduce_type != OCOCC_REDUCE_MEAN || counts || num_segments == 0,
                "MEAN needs counts");
  if (num_segments == 0) return OCOCC_OK;
  OCOCC_REQUIRE(out, "null out");
  const int64_t total = num_segments * c;
  hipLaunchKernelGGL(fill_f32_kernel, dim3(ococc_grid_1d(total, 256)), dim3(256), 0, stream, out,
                     reduce_type == OCOCC_REDUCE_MAX ? arg : (int32_t*)nullptr, total,
                     reduce_type == OCOCC_REDUCE_MAX ? -INFINITY : 0.f);
  OCOCC_CHECK_LAUNCH();
  if (n > 0) {
    OCOCC_REQUIRE(feats && inv, "null feats/inv");
    if (num_segments * 4 > n) {  // fewer than 4 members per segment on average
      const int g2 = ococc_grid_1d(n * c, 256, 8192);
      if (reduce_type == OCOCC_REDUCE_MAX)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(segment_scatter_elem_kernel<OCOCC_REDUCE_MAX>), dim3(g2), dim3(256), 0,
                           stream, feats, inv, n, (int)c, out, counts);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(segment_scatter_elem_kernel<OCOCC_REDUCE_SUM>), dim3(g2), dim3(256), 0,
                           stream, feats, inv, n, (int)c, out, counts);
      OCOCC_CHECK_LAUNCH();
    } else {
    const int cp = lanes_per_row(c);
    // enough waves to cover the chip even for a few thousand rows; long runs per wave otherwise
    int64_t rpw = ococc_cdiv(n * (64 / cp), 8192);
    rpw = rpw < 4 ? 4 : (rpw > kMaxRowsPerWave ? kMaxRowsPerWave : rpw);
    rpw = ococc_align_up(rpw, 64 / cp);
    const int kRowsPerWave = (int)rpw;
    const int grid = ococc_grid_1d(ococc_cdiv(n, kRowsPerWave) * 64, 256);
    if (reduce_type == OCOCC_REDUCE_MAX)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(segment_reduce_kernel<OCOCC_REDUCE_MAX>), dim3(grid),
                         dim3(256), 0, stream, feats, inv, n, (int)c, cp, kRowsPerWave, out, counts);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(segment_reduce_kernel<OCOCC_REDUCE_SUM>), dim3(grid),
                         dim3(256), 0, stream, feats, inv, n, (int)c, cp, kRowsPerWave, out, counts);
    OCOCC_CHECK_LAUNCH();
    }
  }
  if (reduce_type == OCOCC_REDUCE_MEAN) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(finalize_kernel<OCOCC_REDUCE_MEAN>),
                       dim3(ococc_grid_1d(total, 256)), dim3(256), 0, stream, out, counts,
                       num_segments, (int)c);
    OCOCC_CHECK_LAUNCH();
  } else if (reduce_type == OCOCC_REDUCE_MAX) {
    if (arg && n > 0) {
      hipLaunchKernelGGL(argmax_kernel, dim3(ococc_grid_1d(n * c, 256)), dim3(256), 0, stream,
                         feats, inv, n, (int)c, out, arg);
      OCOCC_CHECK_LAUNCH();
    }
    if (counts) {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(finalize_kernel<OCOCC_REDUCE_MAX>),
                         dim3(ococc_grid_1d(total, 256)), dim3(256), 0, stream, out, counts,
                         num_segments, (int)c);
      OCOCC_CHECK_LAUNCH();
    }
  }
  return OCOCC_OK;
}

extern "C" int ococc_segment_reduce_bwd_f32(const float* grad_out, const int32_t* inv, int64_t n,
                                            int32_t c, int32_t reduce_type, const int32_t* counts,
                                            const int32_t* arg, float* grad_feats,
                                            int64_t num_segments, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(n >= 0 && c >= 1, "bad sizes");
  OCOCC_REQUIRE(reduce_type >= 0 && reduce_type <= 2, "reduce_type must be SUM/MEAN/MAX");
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(inv && grad_feats, "null inv/grad_feats");
  OCOCC_REQUIRE(grad_out || num_segments == 0, "null grad_out");
  OCOCC_REQUIRE(reduce_type != OCOCC_REDUCE_MEAN || counts, "MEAN needs counts");
  OCOCC_REQUIRE(reduce_type != OCOCC_REDUCE_MAX || arg, "MAX needs arg");
  const int grid = ococc_grid_1d(n * c, 256);
  if (reduce_type == OCOCC_REDUCE_SUM)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(segment_reduce_bwd_kernel<OCOCC_REDUCE_SUM>), dim3(grid),
                       dim3(256), 0, stream, grad_out, inv, n, (int)c, counts, arg, grad_feats);
  else if (reduce_type == OCOCC_REDUCE_MEAN)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(segment_reduce_bwd_kernel<OCOCC_REDUCE_MEAN>), dim3(grid),
                       dim3(256), 0, stream, grad_out, inv, n, (int)c, counts, arg, grad_feats);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(segment_reduce_bwd_kernel<OCOCC_REDUCE_MAX>), dim3(grid),
                       dim3(256), 0, stream, grad_out, inv, n, (int)c, counts, arg, grad_feats);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
