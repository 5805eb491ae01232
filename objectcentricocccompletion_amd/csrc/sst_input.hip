// B6  element-wise halves of the SST input layer, one launch each where the reference (and round 2's mirror) ran 20-50
// torch operators: the configs[4] step spent its first 3 ms queueing ~250 launches of a few microseconds.
//   ococc_sst_window_coors_i64   get_window_coors for BOTH shifts          mmdet3d/ops/sst/sst_ops.py:266-313
//   ococc_sst_drop_level_i64     drop_single_shift's level / keep decision  mmdet3d/models/middle_encoders/sst_input_layer_v2.py:128-148
//   ococc_sst_pos_embed          get_pos_embed in flat token order          sst_input_layer_v2.py:239-305
// Integer results are exact; the embedding divides by the SAME host-computed frequency table the reference builds with
// torch.pow and calls the same device sinf / cosf.
#include "common.hpp"

namespace {

struct WinGeo {
  int32_t win[3];     // window shape x, y, z
  int32_t maxw[3];    // windows per axis + 1: max_x, max_y, max_z of the reference
  int32_t shift[2][3];
};

__global__ void __launch_bounds__(256)
sst_window_coors_kernel(const int64_t* __restrict__ coors, int64_t n, WinGeo g, int64_t* __restrict__ win_ids,
                        int64_t* __restrict__ in_win) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t b = coors[4 * i], z = coors[4 * i + 1], y = coors[4 * i + 2], x = coors[4 * i + 3];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int64_t sx = x + g.shift[s][0], sy = y + g.shift[s][1], sz = z + g.shift[s][2];
    // (floor division / modulo of torch's integer operators; the shifted coordinates are >= 0)
    const int64_t wx = sx / g.win[0], wy = sy / g.win[1], wz = sz / g.win[2];
    win_ids[s * n + i] = b * ((int64_t)g.maxw[0] * g.maxw[1] * g.maxw[2]) + wx * ((int64_t)g.maxw[1] * g.maxw[2]) +
                         wy * g.maxw[2] + wz;
    int64_t* o = in_win + (s * n + i) * 3;
    o[0] = sz % g.win[2];
    o[1] = sy % g.win[1];
    o[2] = sx % g.win[0];
  }
}

constexpr int kMaxLevels = 8;
struct DropTable {
  int64_t lower[kMaxLevels], upper[kMaxLevels], max_tokens[kMaxLevels], level[kMaxLevels];
  int32_t count;
};
// num_per_voxel = counts[conti]; later table rows win where ranges overlap (the reference assigns in dict order)
__global__ void __launch_bounds__(256)
sst_drop_level_kernel(const int32_t* __restrict__ conti, const int32_t* __restrict__ inner,
                      const int32_t* __restrict__ counts, int64_t n, DropTable t, uint8_t* __restrict__ keep,
                      int64_t* __restrict__ level) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t pop = counts[conti[i]];
  int64_t target = 0, lvl = -1;
  for (int r = 0; r < t.count; ++r)
    if (pop >= t.lower[r] && pop < t.upper[r]) {
      target = t.max_tokens[r];
      lvl = t.level[r];
    }
  keep[i] = (int64_t)inner[i] < target ? 1 : 0;
  level[i] = lvl;
}

struct PosArgs {
  const int64_t* in_win;    // [n, 3] (z, y, x)
  const float* inv_freq;    // [pos_length]
  void* out;                // [n, feat_dim] f32 or bf16
  int64_t n;
  float half[3];            // win_x / 2, win_y / 2, win_z / 2
  float norm[3];            // normalize_pos: 2 * 3.1415 / win (x, y, z), else 0
  int32_t ndim, pos_length, feat_dim, out_bf16;
};
// value of channel j of an axis at in-window coordinate ``coord``
__device__ __forceinline__ float pos_value(const PosArgs& a, int axis, float coord, int j) {
  float p = coord - a.half[axis];
  if (a.norm[axis] != 0.f) p = p / (float)(2.f * a.half[axis]) * 2.f * 3.1415f;
  const float e = p / a.inv_freq[j];
  // stack([e[:, ::2].sin(), e[:, 1::2].cos()], -1).flatten(1): column 2 m is sin(e[2 m]), column 2 m + 1 cos(e[2 m + 1])
  return (j & 1) ? cosf(e) : sinf(e);
}

__global__ void __launch_bounds__(256)
sst_pos_embed_kernel(PosArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n * a.feat_dim) return;
  const int64_t row = i / a.feat_dim;
  const int c = (int)(i - row * a.feat_dim);
  float v = 0.f;
  const int axis = c / a.pos_length;   // 0: x, 1: y, 2: z -- the reference concatenates [x | y | z]
  if (axis < a.ndim) {
    const int j = c - axis * a.pos_length;
    // in_win columns are (z, y, x)
    v = pos_value(a, axis, (float)a.in_win[row * 3 + (2 - axis)], j);
  }
  if (a.out_bf16)
    ((uint16_t*)a.out)[i] = ococc_f32_to_bf16(v);
  else
    ((float*)a.out)[i] = v;
}

// The same values from a table: an in-window coordinate takes ``extent`` values per axis, so the embedding has
// ndim x extent x pos_length distinct numbers (1 008 for 8 x 8 x 8 windows and 128 channels) -- every workgroup works
// them out once (the same expression: same bits) and then only copies; one full-precision sine or cosine per OUTPUT
// element made this launch 110 us for 33 M elements (0.6 TB/s of stores).
constexpr int kPosRows = 256;   // rows per workgroup
constexpr int kPosMaxFeat = 512;
__global__ void __launch_bounds__(256)
sst_pos_embed_table_kernel(PosArgs a, int extent) {
  extern __shared__ float pos_tab[];   // [ndim][extent][pos_length]
  __shared__ int16_t chan_base[kPosMaxFeat];   // channel -> index of its (axis, j) at coordinate 0, -1 for the zero tail
  const int per_axis = extent * a.pos_length;
  for (int t = threadIdx.x; t < a.ndim * per_axis; t += 256) {
    const int axis = t / per_axis, r = t - axis * per_axis, coord = r / a.pos_length;
    pos_tab[t] = pos_value(a, axis, (float)coord, r - coord * a.pos_length);
  }
  for (int c = threadIdx.x; c < a.feat_dim; c += 256) {
    const int axis = c / a.pos_length;
    chan_base[c] = axis < a.ndim ? (int16_t)(axis * per_axis + (c - axis * a.pos_length)) : (int16_t)-1;
  }
  __syncthreads();
  // one item = 8 consecutive channels of a row: three coordinate loads, eight table reads, one 16-byte (bf16) store
  const int chunks = a.feat_dim / 8;
  const int64_t row0 = (int64_t)blockIdx.x * kPosRows;
  // (1 024 rows per workgroup: 106 us, too few loads in flight; the coordinates of four items asked for together: 72 us)
  for (int it = threadIdx.x; it < kPosRows * chunks; it += 256) {
    const int lr = it / chunks, c0 = (it - lr * chunks) * 8;
    const int64_t row = row0 + lr;
    if (row >= a.n) break;
    int64_t co[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) co[d] = a.in_win[row * 3 + d];   // (z, y, x)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int base = chan_base[c0 + u];
      v[u] = 0.f;
      if (base >= 0) {
        const int axis = base / per_axis;
        const int64_t coord = co[2 - axis];
        v[u] = (uint64_t)coord < (uint64_t)extent ? pos_tab[base + (int)coord * a.pos_length]
                                                  : pos_value(a, axis, (float)coord, base - axis * per_axis);
      }
    }
    const int64_t i = row * a.feat_dim + c0;
    if (a.out_bf16) {
      uint4 q;
      q.x = (uint32_t)ococc_f32_to_bf16(v[0]) | ((uint32_t)ococc_f32_to_bf16(v[1]) << 16);
      q.y = (uint32_t)ococc_f32_to_bf16(v[2]) | ((uint32_t)ococc_f32_to_bf16(v[3]) << 16);
      q.z = (uint32_t)ococc_f32_to_bf16(v[4]) | ((uint32_t)ococc_f32_to_bf16(v[5]) << 16);
      q.w = (uint32_t)ococc_f32_to_bf16(v[6]) | ((uint32_t)ococc_f32_to_bf16(v[7]) << 16);
      *(uint4*)((uint16_t*)a.out + i) = q;
    } else {
      *(float4*)((float*)a.out + i) = make_float4(v[0], v[1], v[2], v[3]);
      *(float4*)((float*)a.out + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

}  // namespace

extern "C" int ococc_sst_window_coors_i64(const int64_t* coors, int64_t n, const int32_t* sparse_shape_xyz,
                                          const int32_t* window_shape_xyz, int64_t* win_ids, int64_t* coors_in_win,
                                          ococc_stream_t stream) {
  OCOCC_REQUIRE(n >= 0 && sparse_shape_xyz && window_shape_xyz, "bad arguments");
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(coors && win_ids && coors_in_win, "null pointer");
  WinGeo g;
  for (int d = 0; d < 3; ++d) {
    OCOCC_REQUIRE(window_shape_xyz[d] > 0 && sparse_shape_xyz[d] > 0, "shapes must be positive");
    g.win[d] = window_shape_xyz[d];
    g.maxw[d] = (sparse_shape_xyz[d] + window_shape_xyz[d] - 1) / window_shape_xyz[d] + 1;
    g.shift[0][d] = window_shape_xyz[d];
    g.shift[1][d] = window_shape_xyz[d] / 2;
  }
  if (sparse_shape_xyz[2] == window_shape_xyz[2]) g.shift[0][2] = g.shift[1][2] = 0;   // (2-D windows: no z shift)
  hipLaunchKernelGGL(sst_window_coors_kernel, dim3((unsigned)ococc_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, coors, n,
                     g, win_ids, coors_in_win);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_sst_drop_level_i64(const int32_t* conti, const int32_t* inner, const int32_t* counts, int64_t n,
                                        int32_t num_levels, const int64_t* lower, const int64_t* upper,
                                        const int64_t* max_tokens, const int64_t* level_ids, uint8_t* keep,
                                        int64_t* level, ococc_stream_t stream) {
  OCOCC_REQUIRE(n >= 0 && num_levels >= 0 && num_levels <= kMaxLevels, "at most 8 drop levels");
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(conti && inner && counts && keep && level && (num_levels == 0 || (lower && upper && max_tokens && level_ids)),
                "null pointer");
  DropTable t;
  t.count = num_levels;
  for (int r = 0; r < num_levels; ++r) {
    t.lower[r] = lower[r];
    t.upper[r] = upper[r];
    t.max_tokens[r] = max_tokens[r];
    t.level[r] = level_ids[r];
  }
  hipLaunchKernelGGL(sst_drop_level_kernel, dim3((unsigned)ococc_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, conti,
                     inner, counts, n, t, keep, level);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_sst_pos_embed(const int64_t* coors_in_win, int64_t n, const int32_t* window_shape_xyz, int32_t ndim,
                                   int32_t normalize_pos, const float* inv_freq, int32_t pos_length, int32_t feat_dim,
                                   void* out, int32_t out_dtype, ococc_stream_t stream) {
  OCOCC_REQUIRE(n >= 0 && (ndim == 2 || ndim == 3) && pos_length > 0 && feat_dim >= ndim * pos_length, "bad arguments");
  OCOCC_REQUIRE(out_dtype == OCOCC_BF16 || out_dtype == OCOCC_F32, "out_dtype must be f32/bf16");
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(coors_in_win && inv_freq && out && window_shape_xyz, "null pointer");
  PosArgs a;
  a.in_win = coors_in_win;
  a.inv_freq = inv_freq;
  a.out = out;
  a.n = n;
  for (int d = 0; d < 3; ++d) {
    const int w = (d == 2 && ndim == 2) ? 0 : window_shape_xyz[d];
    a.half[d] = (float)w / 2.f;
    a.norm[d] = normalize_pos ? 1.f : 0.f;
  }
  a.ndim = ndim;
  a.pos_length = pos_length;
  a.feat_dim = feat_dim;
  a.out_bf16 = out_dtype == OCOCC_BF16;
  int extent = 1;
  for (int d = 0; d < ndim; ++d) extent = window_shape_xyz[d] > extent ? window_shape_xyz[d] : extent;
  const int64_t tab_bytes = (int64_t)ndim * extent * pos_length * 4;
  if (tab_bytes <= 32 * 1024 && n >= kPosRows && feat_dim % 8 == 0 && feat_dim <= kPosMaxFeat && ((uintptr_t)out & 15) == 0) {
    hipLaunchKernelGGL(sst_pos_embed_table_kernel, dim3((unsigned)ococc_cdiv(n, kPosRows)), dim3(256), (size_t)tab_bytes,
                       (hipStream_t)stream, a, extent);
  } else {
    hipLaunchKernelGGL(sst_pos_embed_kernel, dim3((unsigned)ococc_cdiv(n * feat_dim, 256)), dim3(256), 0, (hipStream_t)stream,
                       a);
  }
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
