// Tile bodies of the per-point layer kernels (csrc/point_mlp.hip holds the launches of one block, csrc/sir_fused_impl.hpp
// the launches of a whole SIRLayer): device code only, in an anonymous namespace -- every translation unit that includes
// this file gets its own copy.
#pragma once
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

// one tile of work inside a launch: its index (the LayerNorm partial row it owns), its first row, the workgroup's LDS
struct Tile {
  int64_t index, row0;
  float* smem;
  int tid;   // threadIdx.x -- handed in, so that a kernel that runs several bodies in a row can keep the compiler from
             // sharing (and spilling) the per-lane addresses of one body with the next (csrc/sir_fused_impl.hpp)
};

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {   // destination starts at -inf
  const unsigned int bits = __float_as_uint(v);
  if (!(bits >> 31)) atomicMax((int*)addr, (int)bits);
  else atomicMin((unsigned int*)addr, bits);
}

// A value another workgroup of the SAME launch produced with device-scope atomics (the segment maxima, the gradient they
// collect: csrc/sir_fused_impl.hpp): read past the caches that are not coherent between the dies.  COH = false: a plain load.
template <bool COH>
__device__ __forceinline__ float load_shared_result(const float* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

// Padded widths: the k range of a tile is padded with zeros to a multiple of 32 (eight MFMA k-steps: the fragment
// prefetch below needs no bound checks), output channels to a multiple of 16.
__host__ __device__ constexpr int pad_k(int k) { return (k + 31) & ~31; }
__host__ __device__ constexpr int pad_n(int n) { return (n + 15) & ~15; }

// x tile -> LDS.  Wave w builds rows RW w .. RW w + RW - 1 (RW = a quarter of the tile); the 64 lanes of a wave read consecutive columns of one row
// (coalesced).  A lane's columns (lane, lane + 64, ...) keep their source for all rows, so the source pointers are
// worked out once and the row loop is branch-free straight-line code: the loads of four rows are in flight together.
// inv of the wave's rows sits in lanes 0..RW-1 and is handed out by shuffles.
template <int MB, bool COH = false>
__device__ __forceinline__ void assemble(const PointMlpIn& in, int64_t row0, int kp, int ld, float* xs, int* inv_s, int tid) {
  constexpr int RW = 4 * MB;   // rows a wave builds
  const int lane = tid & 63, wave = tid >> 6;
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
        float val;
        if (COH && by_seg[j]) val = load_shared_result<true>(src[j] + seg * stride[j]);
        else val = src[j][(by_seg[j] ? seg : rc) * stride[j]];
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
                                           int64_t rows, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
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
                                         int ld, f32x4 (&acc)[NBW][MB], int tid) {
  const int lane = tid & 63, c = lane & 15, g = lane >> 4;
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
__device__ __forceinline__ void row_sums(float (&part)[MB], float* red, int tid) {
  constexpr int TRM = 16 * MB;
  const int lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
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
  __device__ __forceinline__ Slice(int n, int tid) {
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4;
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
                                               float* red1, float (&rstd)[MB], int tid) {
  float s[MB], q[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    s[mb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[mb] += sl.live[nb][r] ? z[nb][mb][r] : 0.f;
  }
  row_sums(s, red0, tid);
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
  row_sums(q, red1, tid);
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
template <int NBW, int MB, bool COH = false>
__device__ __forceinline__ void
point_mlp_fwd_tile(const PointMlpIn& in, const float* __restrict__ wf, int n, const float* __restrict__ ln_w,
                   const float* __restrict__ ln_b, float eps, int act, float* __restrict__ y, float* __restrict__ vmax,
                   const Tile& tile) {
  const int k = in.ka + in.kb + in.kv, kp = pad_k(k), np = pad_k(n), ld = (kp > np ? kp : np) + 2;
  constexpr int TRM = 16 * MB;       // rows of this instantiation's tile
  float* xs = tile.smem;             // x, then y
  float* red0 = xs + TRM * ld;
  float* red1 = red0 + 4 * TRM;
  int* inv_s = (int*)(red1 + 4 * TRM);
  const int tid = tile.tid, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int64_t row0 = tile.row0;
  assemble<MB, COH>(in, row0, kp, ld, xs, inv_s, tid);
  __syncthreads();
  const Slice<NBW> sl(n, tid);
  f32x4 z[NBW][MB];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) z[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_f32<NBW, MB>(wf, sl.nb0, (n + 15) >> 4, kp >> 2, xs, ld, z, tid);
  __syncthreads();   // x has been read by every wave: the tile now receives y
  if (ln_w) {
    float rstd[MB];
    layernorm_rows<NBW, MB>(z, sl, n, eps, red0, red1, rstd, tid);
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
  store_rows<MB>(xs, ld, y, n, row0, in.rows, tid);
  if (vmax && tid < n) {   // segment maxima: one thread per channel walks the tile's rows
    int cur = -1;
    float acc = 0.f;
#pragma unroll 8
    for (int r = 0; r < TRM; ++r) {
      const int seg = inv_s[r];
      const float v = xs[r * ld + tid];
      if (seg != cur) {
        if (cur >= 0) atomic_max_f32(vmax + (int64_t)cur * n + tid, acc);
        cur = seg;
        acc = v;
      } else {
        acc = fmaxf(acc, v);
      }
    }
    if (cur >= 0) atomic_max_f32(vmax + (int64_t)cur * n + tid, acc);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward.  dy [rows, n] (may be null), dvmax [segments, n] with arg [segments, n] (the row that holds the maximum; may be
// null).  Writes dz [rows, n] (gradient at the Linear's output), xcat [rows, k] (the assembled input, for dW = dz^T xcat;
// may be null), da / dmul [rows, ka], db [rows, kb] (each may be null), adds into dv [segments, kv] (zeroed by the caller)
// and leaves one row [dgamma(n) | dbeta(n)] of LayerNorm partial sums per tile.
// dy [rows, ldy] (row stride ldy >= n: a column slice of a wider gradient is read in place).
// The gradient of the segment maxima may arrive in two parts that are added on the way in: dvmax [segments, ldvm] (a
// column slice of the concatenated maxima's gradient) and dvmax2 [segments, n] (what the next block's gathered copy
// received); either may be null.
template <int NBW, int KBW, int MB, bool COH = false>
__device__ __forceinline__ void
point_mlp_bwd_tile(const PointMlpIn& in, const float* __restrict__ wf, const float* __restrict__ wtf, int n,
                   const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps, int act,
                   const float* __restrict__ dy, int ldy, const float* __restrict__ dvmax, int ldvm,
                   const float* __restrict__ dvmax2, const int32_t* __restrict__ arg,
                   float* __restrict__ dz_out, float* __restrict__ xcat, float* __restrict__ da, float* __restrict__ dmul,
                   float* __restrict__ db, float* __restrict__ dv, float* __restrict__ ln_partial, const Tile& tile) {
  const int k = in.ka + in.kb + in.kv, kp = pad_k(k), np = pad_k(n), ld = (kp > np ? kp : np) + 2;
  constexpr int TRM = 16 * MB;
  float* xs = tile.smem;           // x, then dz, then dx
  float* red0 = xs + TRM * ld;
  float* red1 = red0 + 4 * TRM;
  int* inv_s = (int*)(red1 + 4 * TRM);
  const int tid = tile.tid, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  const int64_t row0 = tile.row0;
  const bool routed = dvmax != nullptr || dvmax2 != nullptr;
  assemble<MB, COH>(in, row0, kp, ld, xs, inv_s, tid);
  __syncthreads();
  if (xcat) store_rows<MB>(xs, ld, xcat, k, row0, in.rows, tid);
  const Slice<NBW> sl(n, tid);
  f32x4 z[NBW][MB];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) z[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_f32<NBW, MB>(wf, sl.nb0, (n + 15) >> 4, kp >> 2, xs, ld, z, tid);
  __syncthreads();   // x has been read (GEMM, xcat copy): the tile now receives dz
  float rstd[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) rstd[mb] = 1.f;
  if (ln_w) layernorm_rows<NBW, MB>(z, sl, n, eps, red0, red1, rstd, tid);   // z = xhat
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
            float u = dy ? dy[row * ldy + ch + r] : 0.f;
            if (routed && arg[(int64_t)segs[mb] * n + ch + r] == (int32_t)row) {
              const float g1 = dvmax ? dvmax[(int64_t)segs[mb] * ldvm + ch + r] : 0.f;
              u += dvmax2 ? g1 + load_shared_result<COH>(dvmax2 + (int64_t)segs[mb] * n + ch + r) : g1;
            }
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
          ln_partial[tile.index * 2 * n + ch + r] = dg;
          ln_partial[tile.index * 2 * n + n + ch + r] = dbt;
        }
      }
    }
  }
  if (ln_w) {
    row_sums<MB>(s1, red0, tid);
    row_sums<MB>(s2, red1, tid);
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
  for (int i = tid; i < TRM * (nk - n); i += kT) {
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
  store_rows<MB>(xs, ld, dz_out, n, row0, in.rows, tid);
  // dx^T[kk][m] = sum_n W^T[kk][n] dz[m][n]: the wave's KBW blocks of 16 input channels
  const int kblocks = (k + 15) >> 4, kb0 = wave * KBW, kbn = max(0, min(KBW, kblocks - kb0));
  f32x4 gx[KBW][MB];
#pragma unroll
  for (int nb = 0; nb < KBW; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) gx[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_f32<KBW, MB>(wtf, kb0, kblocks, nk >> 2, xs, ld, gx, tid);
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
  if (dv && in.kv > 0 && tid < in.kv) {   // gradient of the gathered segment rows: run-length sums, float atomics
    const int col = in.ka + in.kb + tid;
    int cur = -1;
    float acc = 0.f;
#pragma unroll 8
    for (int r = 0; r < TRM; ++r) {
      const int seg = inv_s[r];
      const float v = xs[r * ld + col];
      if (seg != cur) {
        if (cur >= 0) atomicAdd(dv + (int64_t)cur * in.kv + tid, acc);
        cur = seg;
        acc = v;
      } else {
        acc += v;
      }
    }
    if (cur >= 0) atomicAdd(dv + (int64_t)cur * in.kv + tid, acc);
  }
}

// dW partials of a layer: partial[s] = dz[rows of slice s]^T  x_cat[rows of slice s], one [n, k] matrix per row slice; the
// column sums over the slices ride on the pass's end-of-backward reduction (ococc_layernorm_param_reduce_multi reads
// the slab as [slices][2][n k / 2]).  Replaces a batched library GEMM + sum + remainder GEMM + add: the arithmetic is
// nothing (2 rows n k = 0.5 GFLOP at 8 k rows), the four library calls were ~70 us of host time per layer in a
// host-bound step.  Workgroup = one 64 x 64 tile of the product over one row slice, 32 rows per pass through LDS;
// wave w owns the 32 x 32 quadrant (w >> 1, w & 1) as 2 x 2 v_mfma_f32_16x16x4_f32 tiles.
constexpr int kWgLd = 80;   // LDS row stride in floats: the 4 k-groups of an operand read start 16 banks apart
constexpr int kWgradLdsFloats = 2 * 32 * kWgLd;
__device__ __forceinline__ void point_mlp_wgrad_tile(const float* __restrict__ dz, const float* __restrict__ xc, int64_t rows,
                                                     int n, int k, int64_t rows_per_slice, float* __restrict__ partial,
                                                     int slice, int tile_n, int tile_k, float* __restrict__ lds, int tid_) {
  float* zs = lds;                  // [32][kWgLd]
  float* xs = lds + 32 * kWgLd;     // [32][kWgLd]
  const int tid = tid_, lane = tid & 63, wave = tid >> 6;
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

__host__ __device__ inline int nbw_of(int n) { return (((n + 15) >> 4) + 3) / 4; }

}  // namespace
