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

constexpr int TR = 64;          // rows per tile (MB = 4 row blocks of 16); small inputs run 32-row tiles (MB = 2), see tile_mb()
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

// Padded widths: the k range of a tile is padded with zeros to a multiple of 32 (eight MFMA k-steps: the fragment
// prefetch below needs no bound checks), output channels to a multiple of 16.
__host__ __device__ constexpr int pad_k(int k) { return (k + 31) & ~31; }
__host__ __device__ constexpr int pad_n(int n) { return (n + 15) & ~15; }

// x tile -> LDS.  Wave w builds rows RW w .. RW w + RW - 1 (RW = a quarter of the tile); the 64 lanes of a wave read consecutive columns of one row
// (coalesced).  A lane's columns (lane, lane + 64, ...) keep their source for all rows, so the source pointers are
// worked out once and the row loop is branch-free straight-line code: the loads of four rows are in flight together.
// inv of the wave's rows sits in lanes 0..RW-1 and is handed out by shuffles.
template <int MB>
__device__ __forceinline__ void assemble(const PointMlpIn& in, int64_t row0, int kp, int ld, float* xs, int* inv_s) {
  constexpr int RW = 4 * MB;   // rows a wave builds
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = in.ka + in.kb + in.kv, kab = in.ka + in.kb;
  int my_inv = 0;
  if (in.inv && lane < RW) {
    const int64_t row = row0 + RW * wave + lane;
    my_inv = row < in.rows ? in.inv[row] : -1;
    inv_s[RW * wave + lane] = my_inv;
  }
  const float* src[4];     // element (row or segment) 0 of the lane's column in chunk j; a valid address even when unused
  const float* gate[4];
  int64_t stride[4], gstride[4];
  float scale[4];
  bool by_seg[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = lane + 64 * j;
    src[j] = gate[j] = in.a;
    stride[j] = gstride[j] = 0;
    scale[j] = 0.f;
    by_seg[j] = false;
    if (col < in.ka) {
      src[j] = in.a + col;
      stride[j] = in.lda;
      scale[j] = in.colscale ? in.colscale[col] : 1.f;
      if (in.mul) {
        gate[j] = in.mul + col;
        gstride[j] = in.ldm;
      }
    } else if (col < kab) {
      src[j] = in.b + (col - in.ka);
      stride[j] = in.ldb;
      scale[j] = in.bscale;
    } else if (col < k) {
      src[j] = in.v + (col - kab);
      stride[j] = in.kv;
      scale[j] = 1.f;
      by_seg[j] = true;
    }
  }
  const bool gated = in.mul != nullptr;
#pragma unroll 4
  for (int rr = 0; rr < RW; ++rr) {
    const int r = RW * wave + rr;
    const int64_t row = row0 + r;
    const bool ok = row < in.rows;
    const int64_t rc = ok ? row : 0;
    const int64_t seg = max(__shfl(my_inv, rr, 64), 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (64 * j < kp) {
        float val = src[j][(by_seg[j] ? seg : rc) * stride[j]];
        if (gated) val *= gstride[j] ? gate[j][rc * gstride[j]] : 1.f;
        const int col = lane + 64 * j;
        if (col < kp) xs[r * ld + col] = ok ? val * scale[j] : 0.f;
      }
    }
  }
}

// rows of an LDS tile -> rows of a global [rows, width] tensor, a wave per quarter of the tile, lanes along the columns
template <int MB>
__device__ __forceinline__ void store_rows(const float* ts, int ld, float* __restrict__ dst, int width, int64_t row0,
                                           int64_t rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 4
  for (int rr = 0; rr < 4 * MB; ++rr) {
    const int r = 4 * MB * wave + rr;
    if (row0 + r < rows)
      for (int col = lane; col < width; col += 64) dst[(row0 + r) * width + col] = ts[r * ld + col];
  }
}

// acc[nb][mb] += W[16 (nb0 + nb) .. +16][:] x[16 mb .. +16][:]^T ; wf: fragments [n block][k step][64 lanes], k steps
// padded to a multiple of 8.  Blocks beyond the matrix's `nblocks` re-read its last block (their results are dropped).
//
// The weight fragments of a chunk of 8 k-steps are requested while the previous chunk's 8 * NBW * 4 MFMAs run.  The
// compiler will not keep such a prefetch: it sinks plain loads of read-only memory to their uses (one L2 round trip in
// front of every fourth MFMA, measured 3x slower), and volatile loads are serialised with vmcnt(0).  So the loads are
// inline asm, invisible to the compiler, with hand-placed waits (the scheme of csrc/sparse_conv.hip's stream kernel):
// two register sets take turns, no register with a load in flight is copied, `frag_wait<N>` + `frag_tie` stand in front
// of every use (N = the loads of the OTHER set, issued later; older memory operations complete first), and a final
// vmcnt(0) lets the last, unused prefetch land before the registers are reused.
__device__ __forceinline__ void frag_load(float& dst, const float* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p));
}
template <int N>
__device__ __forceinline__ void frag_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
}
__device__ __forceinline__ void frag_tie(float& v) { asm volatile("" : "+v"(v)); }

template <int NBW, int MB>
__device__ __forceinline__ void gemm_f32(const float* __restrict__ wf, int nb0, int nblocks, int ksteps, const float* xs,
                                         int ld, f32x4 (&acc)[NBW][MB]) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const float* wp[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) wp[nb] = wf + (size_t)min(nb0 + nb, nblocks - 1) * ksteps * 64 + lane;
  float a0[NBW][8], a1[NBW][8];
  const int last = ksteps - 8;
  auto issue = [&](float (&dst)[NBW][8], int ks0) {
    const int at = ks0 < last ? ks0 : last;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int j = 0; j < 8; ++j) frag_load(dst[nb][j], wp[nb] + (size_t)(at + j) * 64);
  };
  auto landed = [&](float (&dst)[NBW][8]) {
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int j = 0; j < 8; ++j) frag_tie(dst[nb][j]);
  };
  const float* xb = xs + c * ld + g;
  auto compute = [&](const float (&a)[NBW][8], int ks0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float b[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) b[mb] = xb[mb * 16 * ld + 4 * (ks0 + j)];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb][j], b[mb], acc[nb][mb], 0, 0, 0);
    }
  };
  issue(a0, 0);
  for (int ks0 = 0; ks0 < ksteps; ks0 += 16) {
    issue(a1, ks0 + 8);
    frag_wait<8 * NBW>();
    landed(a0);
    compute(a0, ks0);
    issue(a0, ks0 + 16);
    frag_wait<8 * NBW>();
    landed(a1);
    if (ks0 + 8 < ksteps) compute(a1, ks0 + 8);
  }
  frag_wait<0>();
  landed(a0);
}

// sum over the channels of each of the lane's MB rows (row mb*16 + c), across lanes and waves; one barrier
template <int MB>
__device__ __forceinline__ void row_sums(float (&part)[MB], float* red) {
  constexpr int TRM = 16 * MB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    float p = part[mb];
    p += __shfl_xor(p, 16, 64);
    p += __shfl_xor(p, 32, 64);
    if (g == 0) red[wave * TRM + mb * 16 + c] = p;
  }
  __syncthreads();
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int t = mb * 16 + c;
    part[mb] = (red[t] + red[TRM + t]) + (red[2 * TRM + t] + red[3 * TRM + t]);
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
template <int NBW, int MB>
__device__ __forceinline__ void layernorm_rows(f32x4 (&z)[NBW][MB], const Slice<NBW>& sl, int n, float eps, float* red0,
                                               float* red1, float (&rstd)[MB]) {
  float s[MB], q[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    s[mb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[mb] += sl.live[nb][r] ? z[nb][mb][r] : 0.f;
  }
  row_sums(s, red0);
  const float inv_n = 1.f / (float)n;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
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
  for (int mb = 0; mb < MB; ++mb) {
    rstd[mb] = rsqrtf(q[mb] * inv_n + eps);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[nb][mb][r] *= rstd[mb];
  }
}

template <int ACT>
__device__ __forceinline__ f32x4 act_f4(const f32x4 v) {
  if (ACT == 1) {
    const ln_f32x2 a = ln_gelu2(ln_f32x2{v[0], v[1]}), b = ln_gelu2(ln_f32x2{v[2], v[3]});
    return f32x4{a.x, a.y, b.x, b.y};
  }
  if (ACT == 2) return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
  return v;
}
__device__ __forceinline__ float act_g(int act, float v) {
  return act == 1 ? ln_gelu_grad2(ln_f32x2{v, v}).x : (act == 2 ? (v > 0.f ? 1.f : 0.f) : 1.f);
}

// one LDS tile [tile rows][max(kp, np) + 2] used in turn for x, y / dz and dx, + the LayerNorm exchange + inv
__host__ __device__ constexpr int lds_floats(int kp, int np, int tr) { return tr * ((kp > np ? kp : np) + 2) + 2 * 4 * tr + tr; }

// ---------------------------------------------------------------------------------------------------------------
template <int NBW, int MB>
__global__ void __launch_bounds__(kT, 2)
point_mlp_fwd_kernel(PointMlpIn in, const float* __restrict__ wf, int n, const float* __restrict__ ln_w,
                     const float* __restrict__ ln_b, float eps, int act, float* __restrict__ y, float* __restrict__ vmax) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int k = in.ka + in.kb + in.kv, kp = pad_k(k), np = pad_k(n), ld = (kp > np ? kp : np) + 2;
  constexpr int TRM = 16 * MB;       // rows of this instantiation's tile
  float* xs = smem_f;                // x, then y
  float* red0 = xs + TRM * ld;
  float* red1 = red0 + 4 * TRM;
  int* inv_s = (int*)(red1 + 4 * TRM);
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TRM;
  assemble<MB>(in, row0, kp, ld, xs, inv_s);
  __syncthreads();
  const Slice<NBW> sl(n);
  f32x4 z[NBW][MB];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) z[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_f32<NBW, MB>(wf, sl.nb0, (n + 15) >> 4, kp >> 2, xs, ld, z);
  __syncthreads();   // x has been read by every wave: the tile now receives y
  if (ln_w) {
    float rstd[MB];
    layernorm_rows<NBW, MB>(z, sl, n, eps, red0, red1, rstd);
  }
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    if (nb < sl.nbn) {
      const int ch = 16 * (sl.nb0 + nb) + 4 * g;
      f32x4 gm = {1.f, 1.f, 1.f, 1.f}, bt = {0.f, 0.f, 0.f, 0.f};
      if (ln_w)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (sl.live[nb][r]) {
            gm[r] = ln_w[ch + r];
            bt[r] = ln_b[ch + r];
          }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const f32x4 pre = z[nb][mb] * gm + bt;
        const f32x4 o = act == 1 ? act_f4<1>(pre) : (act == 2 ? act_f4<2>(pre) : pre);
#pragma unroll
        for (int r = 0; r < 4; ++r) xs[(mb * 16 + c) * ld + ch + r] = sl.live[nb][r] ? o[r] : 0.f;
      }
    }
  }
  __syncthreads();
  store_rows<MB>(xs, ld, y, n, row0, in.rows);
  if (vmax && threadIdx.x < n) {   // segment maxima: one thread per channel walks the tile's rows
    int cur = -1;
    float acc = 0.f;
#pragma unroll 8
    for (int r = 0; r < TRM; ++r) {
      const int seg = inv_s[r];
      const float v = xs[r * ld + threadIdx.x];
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
template <int NBW, int KBW, int MB>
__global__ void __launch_bounds__(kT, 2)
point_mlp_bwd_kernel(PointMlpIn in, const float* __restrict__ wf, const float* __restrict__ wtf, int n,
                     const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps, int act,
                     const float* __restrict__ dy, const float* __restrict__ dvmax, const int32_t* __restrict__ arg,
                     float* __restrict__ dz_out, float* __restrict__ xcat, float* __restrict__ da, float* __restrict__ dmul,
                     float* __restrict__ db, float* __restrict__ dv, float* __restrict__ ln_partial) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int k = in.ka + in.kb + in.kv, kp = pad_k(k), np = pad_k(n), ld = (kp > np ? kp : np) + 2;
  constexpr int TRM = 16 * MB;
  float* xs = smem_f;              // x, then dz, then dx
  float* red0 = xs + TRM * ld;
  float* red1 = red0 + 4 * TRM;
  int* inv_s = (int*)(red1 + 4 * TRM);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TRM;
  assemble<MB>(in, row0, kp, ld, xs, inv_s);
  __syncthreads();
  if (xcat) store_rows<MB>(xs, ld, xcat, k, row0, in.rows);
  const Slice<NBW> sl(n);
  f32x4 z[NBW][MB];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) z[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_f32<NBW, MB>(wf, sl.nb0, (n + 15) >> 4, kp >> 2, xs, ld, z);
  __syncthreads();   // x has been read (GEMM, xcat copy): the tile now receives dz
  float rstd[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) rstd[mb] = 1.f;
  if (ln_w) layernorm_rows<NBW, MB>(z, sl, n, eps, red0, red1, rstd);   // z = xhat
  // d(pre-activation) = (dy + routed dvmax) * act'(pre), LayerNorm parameter sums, then the LayerNorm backward
  f32x4 d[NBW][MB];
  float s1[MB], s2[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) s1[mb] = s2[mb] = 0.f;
  int segs[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) segs[mb] = inv_s[mb * 16 + c];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int ch = 16 * (sl.nb0 + nb) + 4 * g;
    // upstream gradients of the lane's 4 x 4 positions of this block, asked for together
    f32x4 up[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int64_t row = row0 + mb * 16 + c;
      up[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (nb < sl.nbn && row < in.rows) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (sl.live[nb][r]) {
            float u = dy ? dy[row * n + ch + r] : 0.f;
            if (dvmax && arg[(int64_t)segs[mb] * n + ch + r] == (int32_t)row) u += dvmax[(int64_t)segs[mb] * n + ch + r];
            up[mb][r] = u;
          }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool live = sl.live[nb][r];
      const float gm = (ln_w && live) ? ln_w[ch + r] : 1.f, bt = (ln_w && live) ? ln_b[ch + r] : 0.f;
      float dg = 0.f, dbt = 0.f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float xh = z[nb][mb][r];
        const float dpre = up[mb][r] * act_g(act, ln_w ? xh * gm + bt : xh);
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
    row_sums<MB>(s1, red0);
    row_sums<MB>(s2, red1);
    const float inv_n = 1.f / (float)n;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          d[nb][mb][r] = sl.live[nb][r] ? ((d[nb][mb][r] - s1[mb] * inv_n) - z[nb][mb][r] * (s2[mb] * inv_n)) * rstd[mb] : 0.f;
  }
  // dz -> LDS, zero up to pad_k(n) columns: the contraction of the second GEMM runs over that range
  const int nk = pad_k(n);
  for (int i = threadIdx.x; i < TRM * (nk - n); i += kT) {
    const int r = i / (nk - n), col = n + i % (nk - n);
    xs[r * ld + col] = 0.f;
  }
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    if (nb < sl.nbn) {
      const int ch = 16 * (sl.nb0 + nb) + 4 * g;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (sl.live[nb][r]) xs[(mb * 16 + c) * ld + ch + r] = d[nb][mb][r];
    }
  }
  __syncthreads();
  store_rows<MB>(xs, ld, dz_out, n, row0, in.rows);
  // dx^T[kk][m] = sum_n W^T[kk][n] dz[m][n]: the wave's KBW blocks of 16 input channels
  const int kblocks = (k + 15) >> 4, kb0 = wave * KBW, kbn = max(0, min(KBW, kblocks - kb0));
  f32x4 gx[KBW][MB];
#pragma unroll
  for (int nb = 0; nb < KBW; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) gx[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_f32<KBW, MB>(wtf, kb0, kblocks, nk >> 2, xs, ld, gx);
  __syncthreads();   // dz has been read (GEMM, dz_out copy): the tile now receives dx
#pragma unroll
  for (int nb = 0; nb < KBW; ++nb) {
    if (nb < kbn) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = 16 * (kb0 + nb) + 4 * g + r;
          if (col < kp) xs[(mb * 16 + c) * ld + col] = gx[nb][mb][r];
        }
    }
  }
  __syncthreads();
  {   // gradients of the direct parts: a wave per quarter of the tile, lanes along the columns
    const int kab = in.ka + in.kb;
#pragma unroll 4
    for (int rr = 0; rr < 4 * MB; ++rr) {
      const int r = 4 * MB * wave + rr;
      const int64_t row = row0 + r;
      if (row >= in.rows) continue;
      for (int col = lane; col < kab; col += 64) {
        const float gxv = xs[r * ld + col];
        if (col < in.ka) {
          const float cs = in.colscale ? in.colscale[col] : 1.f;
          if (da) da[row * in.ka + col] = gxv * (in.mul ? in.mul[row * in.ldm + col] : 1.f) * cs;
          if (dmul) dmul[row * in.ka + col] = gxv * in.a[row * in.lda + col] * cs;
        } else if (db) {
          db[row * in.kb + (col - in.ka)] = gxv * in.bscale;
        }
      }
    }
  }
  if (dv && in.kv > 0 && threadIdx.x < in.kv) {   // gradient of the gathered segment rows: run-length sums, float atomics
    const int col = in.ka + in.kb + threadIdx.x;
    int cur = -1;
    float acc = 0.f;
#pragma unroll 8
    for (int r = 0; r < TRM; ++r) {
      const int seg = inv_s[r];
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

// W [n][k] f32 (element strides) -> f32 MFMA A-operand fragments [ceil(n/16)][pad_k(k)/4][64]: lane 16 g + r of block
// (nb, ks) holds W[16 nb + r][4 ks + g]; zero where the matrix ends
__global__ void __launch_bounds__(256)
point_mlp_pack_kernel(const float* __restrict__ w, int n, int k, int64_t rs, int64_t cs, float* __restrict__ dst) {
  const int ksteps = pad_k(k) >> 2, nblocks = (n + 15) >> 4, total = nblocks * ksteps * 64;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int lane = i & 63, blk = i >> 6, ks = blk % ksteps, nb = blk / ksteps;
    const int row = 16 * nb + (lane & 15), col = 4 * ks + (lane >> 4);
    dst[i] = (row < n && col < k) ? w[row * rs + col * cs] : 0.f;
  }
}

// the same for up to kMaxPack matrices in one launch (a SIR layer packs its four or five weights, both orientations, once
// per optimizer step: 120 launches of 4 us per step otherwise)
constexpr int kMaxPack = 32;
struct PackMulti {
  const float* w[kMaxPack];
  float* dst[kMaxPack];
  int64_t rs[kMaxPack], cs[kMaxPack];
  int32_t n[kMaxPack], k[kMaxPack], first_block[kMaxPack + 1];
  int32_t count;
};
__global__ void __launch_bounds__(256) point_mlp_pack_multi_kernel(PackMulti pk) {
  int t = 0;
  while (t + 1 < pk.count && (int)blockIdx.x >= pk.first_block[t + 1]) ++t;
  const int n = pk.n[t], k = pk.k[t];
  const int ksteps = pad_k(k) >> 2, nblocks = (n + 15) >> 4, total = nblocks * ksteps * 64;
  const int i = ((int)blockIdx.x - pk.first_block[t]) * 256 + threadIdx.x;
  if (i >= total) return;
  const int lane = i & 63, blk = i >> 6, ks = blk % ksteps, nb = blk / ksteps;
  const int row = 16 * nb + (lane & 15), col = 4 * ks + (lane >> 4);
  pk.dst[t][i] = (row < n && col < k) ? pk.w[t][row * pk.rs[t] + col * pk.cs[t]] : 0.f;
}

// arg[seg][ch] = smallest row whose y equals the segment maximum (the rule of the reference's own DynamicScatter,
// scatter_points_cuda.cu:136-160; torch_scatter's tie rule is unpinned): rows are sorted by segment, so the
// first hit of a run is its smallest row; runs of one segment in different tiles meet in an integer atomicMin.
__global__ void __launch_bounds__(256)
segment_argmax_kernel(const float* __restrict__ y, const float* __restrict__ vmax, const int32_t* __restrict__ inv,
                      int64_t rows, int n, int32_t* __restrict__ arg) {
  // one thread per (row, 4 channels): a row that attains its segment's maximum in a channel proposes itself; the smallest
  // proposal stands.  (Few rows per segment attain it: the atomics are rare and spread over all segments.)
  const int q = (n + 3) >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * q) return;
  const int64_t row = i / q;
  const int c0 = (int)(i - row * q) * 4;
  const int seg = inv[row];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int ch = c0 + c;
    if (ch < n && y[row * n + ch] == vmax[(int64_t)seg * n + ch]) atomicMin(arg + (int64_t)seg * n + ch, (int32_t)row);
  }
}

// dW partials of a layer: partial[s] = dz[rows of slice s]^T  x_cat[rows of slice s], one [n, k] matrix per row slice; the
// column sums over the slices ride on the pass's end-of-backward reduction (ococc_layernorm_param_reduce_multi reads
// the slab as [slices][2][n k / 2]).  Replaces a batched library GEMM + sum + remainder GEMM + add: the arithmetic is
// nothing (2 rows n k = 0.5 GFLOP at 8 k rows), the four library calls were ~70 us of host time per layer in a
// host-bound step.  Workgroup = one 64 x 64 tile of the product over one row slice, 32 rows per pass through LDS;
// wave w owns the 32 x 32 quadrant (w >> 1, w & 1) as 2 x 2 v_mfma_f32_16x16x4_f32 tiles.
constexpr int kWgLd = 80;   // LDS row stride in floats: the 4 k-groups of an operand read start 16 banks apart
__device__ __forceinline__ void point_mlp_wgrad_tile(const float* __restrict__ dz, const float* __restrict__ xc, int64_t rows,
                                                     int n, int k, int64_t rows_per_slice, float* __restrict__ partial,
                                                     int slice, int tile_n, int tile_k) {
  __shared__ float zs[32 * kWgLd];
  __shared__ float xs[32 * kWgLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = tile_n * 64, k0 = tile_k * 64;
  const int64_t r_lo = (int64_t)slice * rows_per_slice;
  const int64_t r_hi = r_lo + rows_per_slice < rows ? r_lo + rows_per_slice : rows;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int col = tid & 63, rsub = tid >> 6;          // loader: thread -> (column, row mod 4)
  const bool zc_ok = n0 + col < n, xc_ok = k0 + col < k;
  const int l16 = lane & 15, kg = lane >> 4;
  const int nq = (wave >> 1) * 32, kq = (wave & 1) * 32;
  // the rows of pass i + 1 are requested before the products of pass i: a slice is a handful of passes, and with the
  // loads in front of each pass's barrier the kernel was one global round trip per pass (13.6 us for 8 k rows)
  float zv[8], xv[8];
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t r = r0 + rsub + 4 * j;
      const bool ok = r < r_hi;
      zv[j] = (ok && zc_ok) ? dz[r * n + n0 + col] : 0.f;
      xv[j] = (ok && xc_ok) ? xc[r * k + k0 + col] : 0.f;
    }
  };
  if (r_lo < r_hi) fetch(r_lo);
  for (int64_t r0 = r_lo; r0 < r_hi; r0 += 32) {
    __syncthreads();   // the previous pass has read its tile
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      zs[(rsub + 4 * j) * kWgLd + col] = zv[j];
      xs[(rsub + 4 * j) * kWgLd + col] = xv[j];
    }
    __syncthreads();
    if (r0 + 32 < r_hi) fetch(r0 + 32);
#pragma unroll
    for (int kk = 0; kk < 32; kk += 4) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = zs[(kk + kg) * kWgLd + nq + 16 * i + l16];   // A[m = out channel][contraction row]
        b[i] = xs[(kk + kg) * kWgLd + kq + 16 * i + l16];   // B[contraction row][input channel]
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  float* out = partial + (int64_t)slice * n * k;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = n0 + nq + 16 * i + 4 * kg + q, in = k0 + kq + 16 * j + l16;
        if (ch < n && in < k) out[(int64_t)ch * k + in] = acc[i][j][q];
      }
}

__global__ void __launch_bounds__(256)
point_mlp_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ xc, int64_t rows, int n, int k,
                       int64_t rows_per_slice, float* __restrict__ partial) {
  point_mlp_wgrad_tile(dz, xc, rows, n, k, rows_per_slice, partial, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// the products of SEVERAL layers over the same rows in one launch (the five layers of a SIR layer's backward: 60 of the
// 1.5 k launches of a 4-tracklet step were these, 13 us each for a dozen workgroups of work)
constexpr int kWgradMulti = 8;
struct PointWgradPack {
  const float* dz[kWgradMulti];
  const float* xc[kWgradMulti];
  float* partial[kWgradMulti];
  int32_t n[kWgradMulti], k[kWgradMulti], first[kWgradMulti + 1];
  int32_t count;
};
__global__ void __launch_bounds__(256)
point_mlp_wgrad_multi_kernel(PointWgradPack pk, int64_t rows, int64_t rows_per_slice, int slices) {
  int j = 0;
  while (j + 1 < pk.count && (int)blockIdx.x >= pk.first[j + 1]) ++j;
  const int local = (int)blockIdx.x - pk.first[j];
  const int tiles_n = (pk.n[j] + 63) / 64;
  const int slice = local % slices, t = local / slices;
  point_mlp_wgrad_tile(pk.dz[j], pk.xc[j], rows, pk.n[j], pk.k[j], rows_per_slice, pk.partial[j], slice, t % tiles_n, t / tiles_n);
}

__global__ void __launch_bounds__(256) fill_kernel(float* p, int64_t count, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) p[i] = v;
}

inline int nbw_of(int n) { return (((n + 15) >> 4) + 3) / 4; }

// Rows per tile.  A tile's work is one long dependent chain (assemble, GEMM, LayerNorm, ... ~40 us forward, ~85 us
// backward at 64 rows), and configs[2]'s own batch (4 tracklets, 8-10 k points) is 128-160 tiles of 64 rows on 256
// CUs.  32-row tiles -- twice the workgroups, half the chain, the weight fragments streamed once more per row from
// L2 -- are faster over the whole range the fused layer is used in (whole configs[2] step, MI355X: 17.4 -> 16.0 ms at 4
// tracklets = 8 k points, 28.9 -> 27.5 at 16, 46.7 -> 44.8 at 32, 84.6 -> 82.6 at 64 = 131 k points); beyond that
// (not measured) the 64-row tile's halved weight traffic is kept.  g_force_tile: tests pin any of the three forms.
int g_force_tile = 0;
inline int tile_mb(int64_t rows) {
  if (g_force_tile == 16) return 1;
  if (g_force_tile == 32) return 2;
  if (g_force_tile == 64) return 4;
  if (rows <= 4096) return 1;   // (one or two tracklets: 16-row tiles, ~1 ms of a 15 ms step in two of two A/B pairs; a wash at 8 k rows)
  return ococc_cdiv(rows, TR) <= 2048 ? 2 : 4;
}

constexpr int kMaxDevices = 64;
inline int current_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
  return d % kMaxDevices;
}

}  // namespace

extern "C" int64_t ococc_point_mlp_fragment_floats(int32_t n, int32_t k) {
  return n <= 0 || k <= 0 ? -1 : (int64_t)((n + 15) >> 4) * (pad_k(k) >> 2) * 64;
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

extern "C" int ococc_point_mlp_pack_multi_f32(int32_t count, const void* const* w, const int32_t* n, const int32_t* k,
                                              const int64_t* row_stride, const int64_t* col_stride, void* const* frag,
                                              ococc_stream_t stream) {
  OCOCC_REQUIRE(count >= 0 && count <= kMaxPack, "at most 32 matrices per call");
  if (count == 0) return OCOCC_OK;
  OCOCC_REQUIRE(w && n && k && row_stride && col_stride && frag, "null pointer table");
  PackMulti pk;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    OCOCC_REQUIRE(w[i] && frag[i] && n[i] > 0 && k[i] > 0, "bad matrix");
    pk.w[i] = (const float*)w[i];
    pk.dst[i] = (float*)frag[i];
    pk.rs[i] = row_stride[i];
    pk.cs[i] = col_stride[i];
    pk.n[i] = n[i];
    pk.k[i] = k[i];
    pk.first_block[i] = blocks;
    blocks += (int)ococc_cdiv(ococc_point_mlp_fragment_floats(n[i], k[i]), 256);
  }
  pk.first_block[count] = blocks;
  pk.count = count;
  hipLaunchKernelGGL(point_mlp_pack_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pk);
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
  const int k = ka + kb + kv, kp = pad_k(k), np = pad_k(n);
  const int mb = tile_mb(rows);
  const int lds = lds_floats(kp, np, 16 * mb) * 4;
  const unsigned grid = (unsigned)ococc_cdiv(rows, 16 * mb);
  const int dev = current_device_slot();
#define OCOCC_PM_FWD(NBW, MB)                                                                                             \
  do {                                                                                                                \
    static int lds_set[kMaxDevices] = {};  /* (the attribute sticks: raised once per instantiation and device) */     \
    if (lds > lds_set[dev]) {                                                                                         \
      OCOCC_HIP(hipFuncSetAttribute((const void*)point_mlp_fwd_kernel<NBW, MB>,                                        \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                \
      lds_set[dev] = lds;                                                                                             \
    }                                                                                                                 \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(point_mlp_fwd_kernel<NBW, MB>), dim3(grid), dim3(kT), lds, stream, in, w_frag, \
                       (int)n, ln_weight, ln_bias, eps, (int)act, y, seg_max);                                        \
  } while (0)
  if (mb == 1) {
    switch (nbw_of(n)) {
      case 1: OCOCC_PM_FWD(1, 1); break;
      case 2: OCOCC_PM_FWD(2, 1); break;
      default: OCOCC_PM_FWD(3, 1); break;
    }
  } else if (mb == 2) {
    switch (nbw_of(n)) {
      case 1: OCOCC_PM_FWD(1, 2); break;
      case 2: OCOCC_PM_FWD(2, 2); break;
      default: OCOCC_PM_FWD(3, 2); break;
    }
  } else {
    switch (nbw_of(n)) {
      case 1: OCOCC_PM_FWD(1, 4); break;
      case 2: OCOCC_PM_FWD(2, 4); break;
      default: OCOCC_PM_FWD(3, 4); break;
    }
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
  const int k = ka + kb + kv, kp = pad_k(k), np = pad_k(n);
  const int mb = tile_mb(rows);
  const int lds = lds_floats(kp, np, 16 * mb) * 4;
  const unsigned grid = (unsigned)ococc_cdiv(rows, 16 * mb);
  const int kbw = (((k + 15) >> 4) + 3) / 4;
  const int dev = current_device_slot();
#define OCOCC_PM_BWD(NBW, KBW, MB)                                                                                            \
  do {                                                                                                                    \
    static int lds_set[kMaxDevices] = {};                                                                                 \
    if (lds > lds_set[dev]) {                                                                                             \
      OCOCC_HIP(hipFuncSetAttribute((const void*)point_mlp_bwd_kernel<NBW, KBW, MB>,                                      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                    \
      lds_set[dev] = lds;                                                                                                 \
    }                                                                                                                     \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(point_mlp_bwd_kernel<NBW, KBW, MB>), dim3(grid), dim3(kT), lds, stream, in, w_frag, \
                       wt_frag, (int)n, ln_weight, ln_bias, eps, (int)act, dy, d_seg_max, seg_arg, dz, x_cat, da, dmul,   \
                       db, dv, ln_partial);                                                                               \
  } while (0)
#define OCOCC_PM_BWD_N(KBW, MB)                 \
  switch (nbw_of(n)) {                          \
    case 1: OCOCC_PM_BWD(1, KBW, MB); break;    \
    case 2: OCOCC_PM_BWD(2, KBW, MB); break;    \
    default: OCOCC_PM_BWD(3, KBW, MB); break;   \
  }
#define OCOCC_PM_BWD_K(MB)                      \
  switch (kbw) {                                \
    case 1: OCOCC_PM_BWD_N(1, MB); break;       \
    case 2: OCOCC_PM_BWD_N(2, MB); break;       \
    case 3: OCOCC_PM_BWD_N(3, MB); break;       \
    default: OCOCC_PM_BWD_N(4, MB); break;      \
  }
  if (mb == 1) {
    OCOCC_PM_BWD_K(1)
  } else if (mb == 2) {
    OCOCC_PM_BWD_K(2)
  } else {
    OCOCC_PM_BWD_K(4)
  }
#undef OCOCC_PM_BWD_K
#undef OCOCC_PM_BWD_N
#undef OCOCC_PM_BWD
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int32_t ococc_point_mlp_wgrad_slices(int64_t rows) {
  if (rows <= 0) return 0;
  int64_t s = rows / 256;
  return (int32_t)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

extern "C" int ococc_point_mlp_wgrad_f32(const float* dz, const float* x_cat, int64_t rows, int32_t n, int32_t k,
                                         float* partial, ococc_stream_t stream_) {
  OCOCC_REQUIRE(rows >= 0 && n >= 1 && k >= 1, "bad sizes");
  if (rows == 0) return OCOCC_OK;
  OCOCC_REQUIRE(dz && x_cat && partial, "null pointer");
  const int slices = ococc_point_mlp_wgrad_slices(rows);
  const int64_t per = ococc_align_up(ococc_cdiv(rows, slices), 32);   // (the last slice may come out short or empty: zeros)
  hipLaunchKernelGGL(point_mlp_wgrad_kernel, dim3(slices, (n + 63) / 64, (k + 63) / 64), dim3(256), 0,
                     (hipStream_t)stream_, dz, x_cat, rows, (int)n, (int)k, per, partial);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_point_mlp_wgrad_multi_f32(int32_t count, const float* const* dz, const float* const* x_cat, int64_t rows,
                                               const int32_t* n, const int32_t* k, float* const* partial,
                                               ococc_stream_t stream_) {
  OCOCC_REQUIRE(count >= 1 && count <= kWgradMulti && rows >= 0, "1..8 layers per call");
  OCOCC_REQUIRE(dz && x_cat && n && k && partial, "null pointer table");
  if (rows == 0) return OCOCC_OK;
  const int slices = ococc_point_mlp_wgrad_slices(rows);
  const int64_t per = ococc_align_up(ococc_cdiv(rows, slices), 32);
  PointWgradPack pk;
  int blocks = 0;
  for (int j = 0; j < count; ++j) {
    OCOCC_REQUIRE(dz[j] && x_cat[j] && partial[j] && n[j] >= 1 && k[j] >= 1, "bad layer");
    pk.dz[j] = dz[j]; pk.xc[j] = x_cat[j]; pk.partial[j] = partial[j]; pk.n[j] = n[j]; pk.k[j] = k[j];
    pk.first[j] = blocks;
    blocks += slices * ((n[j] + 63) / 64) * ((k[j] + 63) / 64);
  }
  pk.first[count] = blocks;
  pk.count = count;
  hipLaunchKernelGGL(point_mlp_wgrad_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, pk, rows, per, slices);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int64_t ococc_point_mlp_tiles(int64_t rows) { return rows < 0 ? -1 : ococc_cdiv(rows, 16 * tile_mb(rows)); }

// tests: rows per tile pinned to 32 or 64 (0: by input size)
extern "C" int ococc_point_mlp_force_tile(int32_t tile_rows) {
  OCOCC_REQUIRE(tile_rows == 0 || tile_rows == 16 || tile_rows == 32 || tile_rows == 64, "0 (automatic), 16, 32 or 64");
  g_force_tile = tile_rows;
  return OCOCC_OK;
}

extern "C" int ococc_point_mlp_segment_argmax(const float* y, const float* seg_max, const int32_t* inv, int64_t rows,
                                              int32_t n, int64_t num_segments, int32_t* seg_arg, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(rows >= 0 && n >= 1 && num_segments >= 0, "bad sizes");
  if (num_segments == 0) return OCOCC_OK;
  OCOCC_REQUIRE(y && seg_max && inv && seg_arg, "null pointer");
  OCOCC_HIP(hipMemsetAsync(seg_arg, 0x7f, (size_t)num_segments * n * 4, stream));
  if (rows == 0) return OCOCC_OK;
  hipLaunchKernelGGL(segment_argmax_kernel, dim3((unsigned)ococc_cdiv(rows * ((n + 3) / 4), 256)), dim3(256), 0, stream, y,
                     seg_max, inv, rows, (int)n, seg_arg);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
