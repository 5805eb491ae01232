// A6, fused: one per-point layer of the SIR encoders,  y = act(LN(W x)),  with the input row assembled on the fly and
// the segment maximum of y taken in the same launch.  Replaces, per layer of SIRLayer.forward
// (mmdet3d/models/voxel_encoders/voxel_encoder.py:764-832) and of build_mlp's rel_mlp (mmdet3d/ops/sst/sst_ops.py:333-360):
//   torch.cat / the element-wise products that build the layer input   (voxel_encoder.py:779-805, 818-820)
//   nn.Linear(bias=False) -> LayerNorm(eps 1e-3) -> GELU               (voxel_encoders/utils.py:174-189)
//   scatter_v2(mode='max') = torch_scatter.scatter_max                 (sst_ops.py:150-181)
//   voxel feature back to points: voxel_feats[unq_inv]                 (voxel_encoder.py:756-758, 818)
// The input of a row is  x = [ a (*) mul (*) colscale | b * bscale | v[inv] ]  (any part may be absent):
//   rel_mlp layer 0:  a = f_cluster, colscale = 1 / rel_dist_scaler
//   vfe layer 0:      a = point features, mul = rel_mlp output, colscale = (1 / xyz_normalizer, 1, 1, ...);  b = f_cluster
//                     (with_cluster_center: b * 1 / 10)
//   vfe layer 1:      a = point features of layer 0, v = their segment maxima, inv = the point's segment
// Everything is f32, as the reference computes these layers (force_fp32, voxel_encoder.py:764): the GEMM runs on the
// f32 matrix instruction v_mfma_f32_16x16x4_f32 (157 TFLOP/s peak; the 12 layers of configs[2] are 0.16 TFLOP).
// A workgroup owns 64 consecutive rows: x is built once in LDS ([64][K + 2] floats: the +2 keeps the 16 rows of an MFMA
// operand read on different banks), out^T = W x^T with the weight as the A operand (16 output channels x 4 k, fragment
// order, streamed from L2 eight k-steps ahead) and the rows as the B operand, so a lane ends with 4 consecutive
// channels of one row.  The 4 waves split the output channels; LayerNorm sums cross the waves through LDS.
// Rows arrive sorted by segment (the pooling order, csrc/point_pool.hip): the segment maximum is a run-length walk
// over the tile per channel and one integer atomic per (run, channel) -- exact and order independent.
// Backward (point_mlp_bwd_kernel) recomputes the layer from the same inputs, routes the gradient of the segment maxima
// to the arg-max rows (smallest row index among equals, ococc_segment_argmax), and returns dz (for dW = dz^T x), the
// assembled x (same purpose), the gradients of a, mul and b, and adds the gradient of v with float atomics at run ends.
#include "common.hpp"
#include "ln_math.hpp"

namespace {

constexpr int TR = 64;          // rows per tile
constexpr int kT = 256;         // threads
constexpr int kMaxK = 256, kMaxN = 144;

struct PointMlpIn {
  const float* a;        // [rows, lda], ka columns used
  const float* mul;      // [rows, ldm] or null
  const float* colscale; // [ka] or null
  const float* b;        // [rows, ldb], kb columns
  const float* v;        // [segments, kv]
  const int32_t* inv;    // [rows] segment of every row (non-decreasing), needed for v / vmax
  int32_t ka, lda, ldm, kb, ldb, kv;
  float bscale;
  int64_t rows;
};

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {   // destination starts at -inf
  const unsigned int bits = __float_as_uint(v);
  if (!(bits >> 31)) atomicMax((int*)addr, (int)bits);
  else atomicMin((unsigned int*)addr, bits);
}

// x tile -> LDS (zero beyond the row count and in the padding columns up to kp)
__device__ __forceinline__ void assemble(const PointMlpIn& in, int64_t row0, int kp, int ld, float* xs, int* inv_s) {
  const int k = in.ka + in.kb + in.kv;
  if (in.inv && threadIdx.x < TR) inv_s[threadIdx.x] = row0 + threadIdx.x < in.rows ? in.inv[row0 + threadIdx.x] : -1;
  for (int i = threadIdx.x; i < TR * kp; i += kT) {
    const int r = i / kp, col = i - r * kp;
    const int64_t row = row0 + r;
    float val = 0.f;
    if (row < in.rows && col < k) {
      if (col < in.ka) {
        val = in.a[row * in.lda + col];
        if (in.mul) val *= in.mul[row * in.ldm + col];
        if (in.colscale) val *= in.colscale[col];
      } else if (col < in.ka + in.kb) {
        val = in.b[row * in.ldb + (col - in.ka)] * in.bscale;
      } else {
        val = in.v[(int64_t)in.inv[row] * in.kv + (col - in.ka - in.kb)];
      }
    }
    xs[r * ld + col] = val;
  }
}

// acc[nb][mb] += W[16 (nb0 + nb) .. +16][:] x[16 mb .. +16][:]^T ; wf: fragments [n block][k step][64 lanes]
template <int NBW>
__device__ __forceinline__ void gemm_f32(const float* __restrict__ wf, int nb0, int nbn, int ksteps, const float* xs, int ld,
                                         f32x4 (&acc)[NBW][4]) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const float* wp = wf + (size_t)nb0 * ksteps * 64 + lane;
  float a[NBW][8], an[NBW][8];
  auto fetch = [&](float (&dst)[NBW][8], int ks0) {
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[nb][j] = (nb < nbn && ks0 + j < ksteps) ? wp[((size_t)nb * ksteps + ks0 + j) * 64] : 0.f;
  };
  fetch(a, 0);
  for (int ks0 = 0; ks0 < ksteps; ks0 += 8) {
    fetch(an, ks0 + 8);   // eight k-steps ahead: their L2 latency runs under this chunk's 8 * NBW * 4 MFMAs
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (ks0 + j < ksteps) {
        float b[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) b[mb] = xs[(mb * 16 + c) * ld + 4 * (ks0 + j) + g];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb][j], b[mb], acc[nb][mb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int j = 0; j < 8; ++j) a[nb][j] = an[nb][j];
  }
}

// sum over the channels of each of the lane's 4 rows (row mb*16 + c), across lanes and waves; one barrier
__device__ __forceinline__ void row_sums(float (&part)[4], float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    float p = part[mb];
    p += __shfl_xor(p, 16, 64);
    p += __shfl_xor(p, 32, 64);
    if (g == 0) red[wave * TR + mb * 16 + c] = p;
  }
  __syncthreads();
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int t = mb * 16 + c;
    part[mb] = (red[t] + red[TR + t]) + (red[2 * TR + t] + red[3 * TR + t]);
  }
}

template <int NBW>
struct Slice {   // the wave's channel blocks and which of the lane's channels are real
  int nb0, nbn;
  bool live[NBW][4];
  __device__ __forceinline__ Slice(int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int nblocks = (n + 15) >> 4;
    nb0 = wave * NBW;
    nbn = max(0, min(NBW, nblocks - nb0));
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) live[nb][r] = nb < nbn && 16 * (nb0 + nb) + 4 * g + r < n;
  }
};

// z -> xhat (LayerNorm statistics over the n real channels, two passes), rstd per row block
template <int NBW>
__device__ __forceinline__ void layernorm_rows(f32x4 (&z)[NBW][4], const Slice<NBW>& sl, int n, float eps, float* red0,
                                               float* red1, float (&rstd)[4]) {
  float s[4], q[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    s[mb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[mb] += sl.live[nb][r] ? z[nb][mb][r] : 0.f;
  }
  row_sums(s, red0);
  const float inv_n = 1.f / (float)n;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const float mean = s[mb] * inv_n;
    q[mb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        z[nb][mb][r] = sl.live[nb][r] ? z[nb][mb][r] - mean : 0.f;
        q[mb] += z[nb][mb][r] * z[nb][mb][r];
      }
  }
  row_sums(q, red1);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    rstd[mb] = rsqrtf(q[mb] * inv_n + eps);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[nb][mb][r] *= rstd[mb];
  }
}

__device__ __forceinline__ float act_f(int act, float v) { return act == 1 ? ln_gelu1(v) : (act == 2 ? fmaxf(v, 0.f) : v); }
__device__ __forceinline__ float act_g(int act, float v) {
  return act == 1 ? ln_gelu_grad2(ln_f32x2{v, v}).x : (act == 2 ? (v > 0.f ? 1.f : 0.f) : 1.f);
}

constexpr int lds_floats(int kp, int np) { return TR * (kp + 2) + TR * (np + 2) + 2 * 4 * TR + TR; }

// ---------------------------------------------------------------------------------------------------------------
template <int NBW>
__global__ void __launch_bounds__(kT, 2)
point_mlp_fwd_kernel(PointMlpIn in, const float* __restrict__ wf, int n, const float* __restrict__ ln_w,
                     const float* __restrict__ ln_b, float eps, int act, float* __restrict__ y, float* __restrict__ vmax) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int k = in.ka + in.kb + in.kv, kp = (k + 3) & ~3, ld = kp + 2, np = (n + 15) & ~15, ldy = np + 2;
  float* xs = smem_f;
  float* ys = xs + TR * ld;
  float* red0 = ys + TR * ldy;
  float* red1 = red0 + 4 * TR;
  int* inv_s = (int*)(red1 + 4 * TR);
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TR;
  assemble(in, row0, kp, ld, xs, inv_s);
  __syncthreads();
  const Slice<NBW> sl(n);
  f32x4 z[NBW][4];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) z[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_f32<NBW>(wf, sl.nb0, sl.nbn, kp >> 2, xs, ld, z);
  if (ln_w) {
    float rstd[4];
    layernorm_rows<NBW>(z, sl, n, eps, red0, red1, rstd);
  }
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    if (nb < sl.nbn) {
      const int ch = 16 * (sl.nb0 + nb) + 4 * g;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = z[nb][mb][r];
          if (ln_w && sl.live[nb][r]) v = v * ln_w[ch + r] + ln_b[ch + r];
          ys[(mb * 16 + c) * ldy + ch + r] = sl.live[nb][r] ? act_f(act, v) : 0.f;
        }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TR * n; i += kT) {   // rows of y, coalesced
    const int r = i / n, col = i - r * n;
    if (row0 + r < in.rows) y[(row0 + r) * n + col] = ys[r * ldy + col];
  }
  if (vmax && threadIdx.x < n) {   // segment maxima: one thread per channel walks the tile's rows
    int cur = -1;
    float acc = 0.f;
    for (int r = 0; r < TR; ++r) {
      const int seg = inv_s[r];
      if (seg < 0) break;
      const float v = ys[r * ldy + threadIdx.x];
      if (seg != cur) {
        if (cur >= 0) atomic_max_f32(vmax + (int64_t)cur * n + threadIdx.x, acc);
        cur = seg;
        acc = v;
      } else {
        acc = fmaxf(acc, v);
      }
    }
    if (cur >= 0) atomic_max_f32(vmax + (int64_t)cur * n + threadIdx.x, acc);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward.  dy [rows, n] (may be null), dvmax [segments, n] with arg [segments, n] (the row that holds the maximum; may be
// null).  Writes dz [rows, n] (gradient at the Linear's output), xcat [rows, k] (the assembled input, for dW = dz^T xcat;
// may be null), da / dmul [rows, ka], db [rows, kb] (each may be null), adds into dv [segments, kv] (zeroed by the caller)
// and leaves one row [dgamma(n) | dbeta(n)] of LayerNorm partial sums per tile.
template <int NBW, int KBW>
__global__ void __launch_bounds__(kT, 2)
point_mlp_bwd_kernel(PointMlpIn in, const float* __restrict__ wf, const float* __restrict__ wtf, int n,
                     const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps, int act,
                     const float* __restrict__ dy, const float* __restrict__ dvmax, const int32_t* __restrict__ arg,
                     float* __restrict__ dz_out, float* __restrict__ xcat, float* __restrict__ da, float* __restrict__ dmul,
                     float* __restrict__ db, float* __restrict__ dv, float* __restrict__ ln_partial) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int k = in.ka + in.kb + in.kv, kp = (k + 3) & ~3, ld = kp + 2, np = (n + 15) & ~15, ldy = np + 2;
  float* xs = smem_f;              // x, then dx
  float* ys = xs + TR * ld;        // dz
  float* red0 = ys + TR * ldy;
  float* red1 = red0 + 4 * TR;
  int* inv_s = (int*)(red1 + 4 * TR);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TR;
  assemble(in, row0, kp, ld, xs, inv_s);
  __syncthreads();
  if (xcat)
    for (int i = threadIdx.x; i < TR * k; i += kT) {
      const int r = i / k, col = i - r * k;
      if (row0 + r < in.rows) xcat[(row0 + r) * k + col] = xs[r * ld + col];
    }
  const Slice<NBW> sl(n);
  f32x4 z[NBW][4];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) z[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_f32<NBW>(wf, sl.nb0, sl.nbn, kp >> 2, xs, ld, z);
  float rstd[4] = {1.f, 1.f, 1.f, 1.f};
  if (ln_w) layernorm_rows<NBW>(z, sl, n, eps, red0, red1, rstd);   // z = xhat
  // d(pre-activation) = (dy + routed dvmax) * act'(pre), LayerNorm parameter sums, then the LayerNorm backward
  f32x4 d[NBW][4];
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int ch = 16 * (sl.nb0 + nb) + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool live = sl.live[nb][r];
      const float gm = (ln_w && live) ? ln_w[ch + r] : 1.f, bt = (ln_w && live) ? ln_b[ch + r] : 0.f;
      float dg = 0.f, dbt = 0.f;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int64_t row = row0 + mb * 16 + c;
        float up = 0.f;
        if (live && row < in.rows) {
          if (dy) up = dy[row * n + ch + r];
          if (dvmax) {
            const int64_t seg = inv_s[mb * 16 + c];
            if (arg[seg * n + ch + r] == (int32_t)row) up += dvmax[seg * n + ch + r];
          }
        }
        const float xh = z[nb][mb][r];
        const float dpre = up * act_g(act, ln_w ? xh * gm + bt : xh);
        dg += dpre * xh;
        dbt += dpre;
        const float v = dpre * gm;
        d[nb][mb][r] = v;
        s1[mb] += v;
        s2[mb] += v * xh;
      }
      if (ln_w && ln_partial) {   // sums over the lane's rows, then over the 16 lanes that hold the other rows
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          dg += __shfl_xor(dg, m, 64);
          dbt += __shfl_xor(dbt, m, 64);
        }
        if (c == 0 && live) {
          ln_partial[(int64_t)blockIdx.x * 2 * n + ch + r] = dg;
          ln_partial[(int64_t)blockIdx.x * 2 * n + n + ch + r] = dbt;
        }
      }
    }
  }
  if (ln_w) {
    row_sums(s1, red0);
    row_sums(s2, red1);
    const float inv_n = 1.f / (float)n;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          d[nb][mb][r] = sl.live[nb][r] ? ((d[nb][mb][r] - s1[mb] * inv_n) - z[nb][mb][r] * (s2[mb] * inv_n)) * rstd[mb] : 0.f;
  }
  // dz -> LDS (zero in the padding channels, so that the contraction below may run over np)
  for (int i = threadIdx.x; i < TR * (np - n); i += kT) {
    const int r = i / (np - n), col = n + i % (np - n);
    ys[r * ldy + col] = 0.f;
  }
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    if (nb < sl.nbn) {
      const int ch = 16 * (sl.nb0 + nb) + 4 * g;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (sl.live[nb][r]) ys[(mb * 16 + c) * ldy + ch + r] = d[nb][mb][r];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TR * n; i += kT) {
    const int r = i / n, col = i - r * n;
    if (row0 + r < in.rows) dz_out[(row0 + r) * n + col] = ys[r * ldy + col];
  }
  // dx^T[kk][m] = sum_n W^T[kk][n] dz[m][n]: the wave's KBW blocks of 16 input channels
  const int kblocks = (kp + 15) >> 4, kb0 = wave * KBW, kbn = max(0, min(KBW, kblocks - kb0));
  f32x4 gx[KBW][4];
#pragma unroll
  for (int nb = 0; nb < KBW; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) gx[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_f32<KBW>(wtf, kb0, kbn, (n + 3) >> 2, ys, ldy, gx);   // (dz is zero in [n, np): the last k-step may run past n)
  __syncthreads();   // every wave has left x (the forward GEMM's operand) and dz behind
#pragma unroll
  for (int nb = 0; nb < KBW; ++nb) {
    if (nb < kbn) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = 16 * (kb0 + nb) + 4 * g + r;
          if (col < kp) xs[(mb * 16 + c) * ld + col] = gx[nb][mb][r];
        }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TR * (in.ka + in.kb); i += kT) {   // gradients of the direct parts
    const int r = i / (in.ka + in.kb), col = i - r * (in.ka + in.kb);
    const int64_t row = row0 + r;
    if (row >= in.rows) continue;
    const float gxv = xs[r * ld + col];
    if (col < in.ka) {
      const float cs = in.colscale ? in.colscale[col] : 1.f;
      const float av = in.a[row * in.lda + col];
      const float mv = in.mul ? in.mul[row * in.ldm + col] : 1.f;
      if (da) da[row * in.ka + col] = gxv * mv * cs;
      if (dmul) dmul[row * in.ka + col] = gxv * av * cs;
    } else if (db) {
      db[row * in.kb + (col - in.ka)] = gxv * in.bscale;
    }
  }
  if (dv && in.kv > 0 && threadIdx.x < in.kv) {   // gradient of the gathered segment rows: run-length sums, float atomics
    const int col = in.ka + in.kb + threadIdx.x;
    int cur = -1;
    float acc = 0.f;
    for (int r = 0; r < TR; ++r) {
      const int seg = inv_s[r];
      if (seg < 0) break;
      const float v = xs[r * ld + col];
      if (seg != cur) {
        if (cur >= 0) atomicAdd(dv + (int64_t)cur * in.kv + threadIdx.x, acc);
        cur = seg;
        acc = v;
      } else {
        acc += v;
      }
    }
    if (cur >= 0) atomicAdd(dv + (int64_t)cur * in.kv + threadIdx.x, acc);
  }
}

// W [n][k] f32 (element strides) -> f32 MFMA A-operand fragments [ceil(n/16)][ceil(k/4)][64]: lane 16 g + r of block
// (nb, ks) holds W[16 nb + r][4 ks + g]; zero where the matrix ends
__global__ void __launch_bounds__(256)
point_mlp_pack_kernel(const float* __restrict__ w, int n, int k, int64_t rs, int64_t cs, float* __restrict__ dst) {
  const int ksteps = (k + 3) >> 2, nblocks = (n + 15) >> 4, total = nblocks * ksteps * 64;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int lane = i & 63, blk = i >> 6, ks = blk % ksteps, nb = blk / ksteps;
    const int row = 16 * nb + (lane & 15), col = 4 * ks + (lane >> 4);
    dst[i] = (row < n && col < k) ? w[row * rs + col * cs] : 0.f;
  }
}

// arg[seg][ch] = smallest row whose y equals the segment maximum (the rule of the reference's own DynamicScatter,
// scatter_points_cuda.cu:136-160; torch_scatter's tie rule is unpinned): rows are sorted by segment, so the
// first hit of a run is its smallest row; runs of one segment in different tiles meet in an integer atomicMin.
__global__ void __launch_bounds__(256)
segment_argmax_kernel(const float* __restrict__ y, const float* __restrict__ vmax, const int32_t* __restrict__ inv,
                      int64_t rows, int n, int32_t* __restrict__ arg) {
  const int64_t row0 = (int64_t)blockIdx.x * TR;
  for (int ch = threadIdx.x; ch < n; ch += 256) {
    int cur = -1;
    float top = 0.f;
    bool found = false;
    for (int r = 0; r < TR && row0 + r < rows; ++r) {
      const int seg = inv[row0 + r];
      if (seg != cur) {
        cur = seg;
        top = vmax[(int64_t)seg * n + ch];
        found = false;
      }
      if (!found && y[(row0 + r) * n + ch] == top) {
        atomicMin(arg + (int64_t)seg * n + ch, (int32_t)(row0 + r));
        found = true;
      }
    }
  }
}

__global__ void __launch_bounds__(256) fill_kernel(float* p, int64_t count, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) p[i] = v;
}

inline int nbw_of(int n) { return (((n + 15) >> 4) + 3) / 4; }

}  // namespace

extern "C" int64_t ococc_point_mlp_fragment_floats(int32_t n, int32_t k) {
  return n <= 0 || k <= 0 ? -1 : (int64_t)((n + 15) >> 4) * ((k + 3) >> 2) * 64;
}

extern "C" int ococc_point_mlp_pack_f32(const float* w, int32_t n, int32_t k, int64_t row_stride, int64_t col_stride,
                                        float* frag, ococc_stream_t stream) {
  OCOCC_REQUIRE(w && frag && n > 0 && k > 0, "bad arguments");
  const int64_t total = ococc_point_mlp_fragment_floats(n, k);
  hipLaunchKernelGGL(point_mlp_pack_kernel, dim3(ococc_grid_1d(total, 256, 256)), dim3(256), 0, (hipStream_t)stream, w, (int)n,
                     (int)k, row_stride, col_stride, frag);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

static int check_in(const PointMlpIn& in, int n) {
  const int k = in.ka + in.kb + in.kv;
  OCOCC_REQUIRE(in.rows >= 0 && in.ka >= 0 && in.kb >= 0 && in.kv >= 0 && k >= 1 && k <= kMaxK && n >= 1 && n <= kMaxN,
                "1 <= input columns <= 256, 1 <= output channels <= 144");
  OCOCC_REQUIRE((in.ka == 0 || in.a) && (in.kb == 0 || in.b) && (in.kv == 0 || (in.v && in.inv)), "null input part");
  OCOCC_REQUIRE(in.lda >= in.ka && in.ldb >= in.kb && (!in.mul || in.ldm >= in.ka), "row strides shorter than the columns");
  return OCOCC_OK;
}

extern "C" int ococc_point_mlp_fwd_f32(const float* a, int32_t ka, int32_t lda, const float* mul, int32_t ldm,
                                       const float* colscale, const float* b, int32_t kb, int32_t ldb, float bscale,
                                       const float* v, int32_t kv, const int32_t* inv, int64_t rows, const float* w_frag,
                                       int32_t n, const float* ln_weight, const float* ln_bias, float eps, int32_t act,
                                       float* y, float* seg_max, int64_t num_segments, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PointMlpIn in{a, mul, colscale, b, v, inv, ka, lda, ldm, kb, ldb, kv, bscale, rows};
  if (int rc = check_in(in, n)) return rc;
  OCOCC_REQUIRE(w_frag && y && (act >= 0 && act <= 2) && (!ln_weight == !ln_bias), "bad arguments");
  OCOCC_REQUIRE(!seg_max || (inv && num_segments >= 0), "segment maxima need inv and the segment count");
  if (seg_max && num_segments > 0)
    hipLaunchKernelGGL(fill_kernel, dim3(ococc_grid_1d(num_segments * n, 256, 1024)), dim3(256), 0, stream, seg_max,
                       num_segments * n, -INFINITY);
  if (rows == 0) return OCOCC_OK;
  const int k = ka + kb + kv, kp = (k + 3) & ~3, np = (n + 15) & ~15;
  const int lds = lds_floats(kp, np) * 4;
  const unsigned grid = (unsigned)ococc_cdiv(rows, TR);
#define OCOCC_PM_FWD(NBW)                                                                                             \
  do {                                                                                                                \
    OCOCC_HIP(hipFuncSetAttribute((const void*)point_mlp_fwd_kernel<NBW>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  lds));                                                                              \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(point_mlp_fwd_kernel<NBW>), dim3(grid), dim3(kT), lds, stream, in, w_frag,     \
                       (int)n, ln_weight, ln_bias, eps, (int)act, y, seg_max);                                        \
  } while (0)
  switch (nbw_of(n)) {
    case 1: OCOCC_PM_FWD(1); break;
    case 2: OCOCC_PM_FWD(2); break;
    default: OCOCC_PM_FWD(3); break;
  }
#undef OCOCC_PM_FWD
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_point_mlp_bwd_f32(const float* a, int32_t ka, int32_t lda, const float* mul, int32_t ldm,
                                       const float* colscale, const float* b, int32_t kb, int32_t ldb, float bscale,
                                       const float* v, int32_t kv, const int32_t* inv, int64_t rows, const float* w_frag,
                                       const float* wt_frag, int32_t n, const float* ln_weight, const float* ln_bias,
                                       float eps, int32_t act, const float* dy, const float* d_seg_max,
                                       const int32_t* seg_arg, float* dz, float* x_cat, float* da, float* dmul, float* db,
                                       float* dv, float* ln_partial, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PointMlpIn in{a, mul, colscale, b, v, inv, ka, lda, ldm, kb, ldb, kv, bscale, rows};
  if (int rc = check_in(in, n)) return rc;
  OCOCC_REQUIRE(w_frag && wt_frag && dz && (act >= 0 && act <= 2) && (!ln_weight == !ln_bias), "bad arguments");
  OCOCC_REQUIRE((!d_seg_max) == (!seg_arg) && (!d_seg_max || inv), "the gradient of the segment maxima comes with seg_arg and inv");
  OCOCC_REQUIRE(!ln_weight || ln_partial, "LayerNorm partial rows missing");
  if (rows == 0) return OCOCC_OK;
  const int k = ka + kb + kv, kp = (k + 3) & ~3, np = (n + 15) & ~15;
  const int lds = lds_floats(kp, np) * 4;
  const unsigned grid = (unsigned)ococc_cdiv(rows, TR);
  const int kbw = (((kp + 15) >> 4) + 3) / 4;
#define OCOCC_PM_BWD(NBW, KBW)                                                                                            \
  do {                                                                                                                    \
    OCOCC_HIP(hipFuncSetAttribute((const void*)point_mlp_bwd_kernel<NBW, KBW>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  lds));                                                                                  \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(point_mlp_bwd_kernel<NBW, KBW>), dim3(grid), dim3(kT), lds, stream, in, w_frag,    \
                       wt_frag, (int)n, ln_weight, ln_bias, eps, (int)act, dy, d_seg_max, seg_arg, dz, x_cat, da, dmul,   \
                       db, dv, ln_partial);                                                                               \
  } while (0)
#define OCOCC_PM_BWD_N(KBW)                     \
  switch (nbw_of(n)) {                          \
    case 1: OCOCC_PM_BWD(1, KBW); break;        \
    case 2: OCOCC_PM_BWD(2, KBW); break;        \
    default: OCOCC_PM_BWD(3, KBW); break;       \
  }
  switch (kbw) {
    case 1: OCOCC_PM_BWD_N(1); break;
    case 2: OCOCC_PM_BWD_N(2); break;
    case 3: OCOCC_PM_BWD_N(3); break;
    default: OCOCC_PM_BWD_N(4); break;
  }
#undef OCOCC_PM_BWD_N
#undef OCOCC_PM_BWD
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int64_t ococc_point_mlp_tiles(int64_t rows) { return rows < 0 ? -1 : ococc_cdiv(rows, TR); }

extern "C" int ococc_point_mlp_segment_argmax(const float* y, const float* seg_max, const int32_t* inv, int64_t rows,
                                              int32_t n, int64_t num_segments, int32_t* seg_arg, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(rows >= 0 && n >= 1 && num_segments >= 0, "bad sizes");
  if (num_segments == 0) return OCOCC_OK;
  OCOCC_REQUIRE(y && seg_max && inv && seg_arg, "null pointer");
  OCOCC_HIP(hipMemsetAsync(seg_arg, 0x7f, (size_t)num_segments * n * 4, stream));
  if (rows == 0) return OCOCC_OK;
  hipLaunchKernelGGL(segment_argmax_kernel, dim3((unsigned)ococc_cdiv(rows, TR)), dim3(256), 0, stream, y, seg_max, inv,
                     rows, (int)n, seg_arg);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
