// B3 sub-manifold rulebook for gfx950.
// Reference: spconv::getIndicePairsSubM (mmdet3d/ops/spconv/include/spconv/
// geometry.h:247-297, CPU) and prepareSubMGridKernel / getSubMIndicePairsKernel
// (include/spconv/indice.cu.h:147-234, GPU; pair order there is decided by
// atomicAdd and is not reproducible).
//
// MI355X design.  The reference fills a dense int32 grid (16 MB for 64 x 40^3)
// on every call.  Here the active set is a 1-bit-per-cell bitmap plus a
// popcount prefix (1 MB, L2 resident): rank(cell) gives the voxel's position
// in sorted order and perm[rank] its row.  The neighbour table
//   nbr_t[k][o] = row of the active voxel at pos(o) + (k - centre), else -1
// is written offset-major so that both this kernel's stores and the
// convolution's index loads are contiguous.  A wave ballot over each table
// row produces, for every 16-row block, the set of offsets that have any
// neighbour (blockmask) -- the convolution skips the rest with scalar
// branches.  The reference-format rulebook (indice_pairs / indice_num) is
// derived by an order preserving stream compaction (prefix sum per offset),
// which reproduces the CPU functor's order exactly: within offset k pairs are
// in ascending input row j, and by symmetry of a sub-manifold
// (in j feeds out o through k  <=>  nbr_t[K-1-k][j] = o).
#include "common.hpp"
#include "scan.hpp"

namespace {

struct Geom {
  int32_t batch, D, H, W;
  int32_t kd, kh, kw;
  // generic path (template KS = 0): table entry k = (kz,ky,kx) of row at q is the row at q + start + k * step.
  // Plain sub-manifold: start = -k/2, step = 1.  Dilated sub-manifold (geometry.h:24-85 with the stride 1 /
  // padding k/2 the reference forces, spconv_ops.h:66-83): gather side start = -k/2, step = +dilation; scatter side
  // start = +k/2, step = -dilation -- NOT mirror images of each other unless dilation = 1.
  int32_t sz, sy, sx, tz, ty, tx;
};

__device__ __forceinline__ int32_t rank_of(const uint32_t* __restrict__ bitmap,
                                           const uint32_t* __restrict__ prefix, int64_t cell) {
  const uint32_t w = bitmap[cell >> 5];
  const uint32_t bit = 1u << (cell & 31);
  if (!(w & bit)) return -1;
  return (int32_t)(prefix[cell >> 5] + __popc(w & (bit - 1u)));
}

__global__ void __launch_bounds__(256)
mark_voxels_kernel(const int32_t* __restrict__ indices, int64_t n, Geom g,
                   uint32_t* __restrict__ bitmap, int32_t* __restrict__ status) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t b = indices[i * 4], z = indices[i * 4 + 1], y = indices[i * 4 + 2],
                  x = indices[i * 4 + 3];
    if ((unsigned)b >= (unsigned)g.batch || (unsigned)z >= (unsigned)g.D ||
        (unsigned)y >= (unsigned)g.H || (unsigned)x >= (unsigned)g.W) {
      if (status) *status = 1;
      continue;
    }
    const int64_t cell = (((int64_t)b * g.D + z) * g.H + y) * g.W + x;
    atomicOr(bitmap + (cell >> 5), 1u << (cell & 31));
  }
}

// perm[rank(cell of row j)] = j + 1  (largest j wins if a cell is listed twice; 0 = empty)
__global__ void __launch_bounds__(256)
fill_perm_kernel(const int32_t* __restrict__ indices, int64_t n, Geom g,
                 const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ prefix,
                 int32_t* __restrict__ perm) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t b = indices[i * 4], z = indices[i * 4 + 1], y = indices[i * 4 + 2],
                  x = indices[i * 4 + 3];
    if ((unsigned)b >= (unsigned)g.batch || (unsigned)z >= (unsigned)g.D ||
        (unsigned)y >= (unsigned)g.H || (unsigned)x >= (unsigned)g.W)
      continue;
    const int64_t cell = (((int64_t)b * g.D + z) * g.H + y) * g.W + x;
    const int32_t r = rank_of(bitmap, prefix, cell);
    if (r >= 0) atomicMax(perm + r, (int32_t)i + 1);  // row + 1, so that a zero fill means empty
  }
}

// thread = output row o, looping over the kvol offsets; the 16-row block masks come out of wave
// ballots and are written once per block (no atomics, so blockmask needs no zero fill)
template <int KS>  // compile-time kernel size (3x3x3 fast path) or 0 = sizes from g
__global__ void __launch_bounds__(256)
neighbour_table_kernel(const int32_t* __restrict__ indices, int64_t n, Geom g,
                       const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ prefix,
                       const int32_t* __restrict__ perm, int32_t* __restrict__ nbr_t,
                       uint32_t* __restrict__ blockmask, uint32_t* __restrict__ blk_cnt) {
  // perm == nullptr: rows are in ascending cell order (the output of ococc_grid_unique_i32), rank = row
  // blk_cnt[k * gridDim.x + block]: valid entries of this 256-row block at offset k (for the compaction)
  __shared__ uint32_t wcnt[4][64];
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int kd = KS ? KS : g.kd, kh = KS ? KS : g.kh, kw = KS ? KS : g.kw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int32_t b = -1, z0 = 0, y0 = 0, x0 = 0;
  if (o < n) {
    b = indices[o * 4];
    z0 = indices[o * 4 + 1] + (KS ? -(KS / 2) : g.sz);
    y0 = indices[o * 4 + 2] + (KS ? -(KS / 2) : g.sy);
    x0 = indices[o * 4 + 3] + (KS ? -(KS / 2) : g.sx);
  }
  const bool row_ok = (unsigned)b < (unsigned)g.batch;
  uint32_t mask = 0;
  // cell of the (0,0,0) corner of the window in 32-bit arithmetic (cells < 2^31, make_layout), then one
  // add per offset; a negative corner coordinate makes base meaningless but those offsets are masked out
  const int32_t base = row_ok ? ((b * g.D + z0) * g.H + y0) * g.W + x0 : 0;
  auto lookup = [&](int kz, int ky, int kx) -> int32_t {
    const int32_t oz = KS ? kz : kz * g.tz, oy = KS ? ky : ky * g.ty, ox = KS ? kx : kx * g.tx;
    const int32_t z = z0 + oz, y = y0 + oy, x = x0 + ox;
    int32_t v = -1;
    if (row_ok && (unsigned)z < (unsigned)g.D && (unsigned)y < (unsigned)g.H &&
        (unsigned)x < (unsigned)g.W) {
      const int32_t cell = base + (oz * g.H + oy) * g.W + ox;
      const int32_t r = rank_of(bitmap, prefix, cell);
      if (r >= 0) v = perm ? perm[r] - 1 : r;  // perm holds row + 1 (0 = empty)
    }
    return v;
  };
  auto emit = [&](int32_t v, int k) {
    if (o < n) nbr_t[(int64_t)k * n + o] = v;
    const unsigned long long m = __ballot(v >= 0);
    if (k < 32 && ((m >> (lane & 48)) & 0xffffull)) mask |= 1u << k;
    if (blk_cnt && lane == 0) {
      if (k < 64) wcnt[wave][k] = (uint32_t)__popcll(m);
      else atomicAdd(blk_cnt + (int64_t)k * gridDim.x + blockIdx.x, (uint32_t)__popcll(m));  // kvol > 64: rare
    }
  };
  if constexpr (KS == 3) {
    // All lookups first (their loads are independent and overlap), then the ballots and stores.  The three
    // x-neighbours of one (kz, ky) are consecutive cells: ONE bitmap word (two when they straddle a word
    // boundary) and ONE prefix word serve all three -- rank(cell) = prefix[word] + popcount of the bits below
    // it, taken over the 64-bit pair -- 9 + 9 loads per voxel instead of 27 + 27.
    int32_t v[27];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int32_t z = z0 + kz, y = y0 + ky;
        const bool zy_ok = row_ok && (unsigned)z < (unsigned)g.D && (unsigned)y < (unsigned)g.H;
        const int32_t c0 = base + (kz * g.H + ky) * g.W;  // cell of kx = 0; -1 only for x0 = -1 at the grid's first cell
        const int32_t cs = c0 < 0 ? 0 : c0;
        const int sh = (cs & 31) - (c0 < 0 ? 1 : 0);      // bit position of the kx = 0 cell in the pair below
        uint32_t w_lo = 0, w_hi = 0;
        if (zy_ok) {
          w_lo = bitmap[cs >> 5];
          // (2 of 32 positions; behind the grid's last word this reads the start of the prefix array, which
          // shares the workspace -- those bits belong to x >= W and are masked by the bound check below)
          if ((cs & 31) > 29) w_hi = bitmap[(cs >> 5) + 1];
        }
        const unsigned long long both = ((unsigned long long)w_hi << 32) | w_lo;
        uint32_t bits3 = 0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int pos = sh + kx;
          if (zy_ok && (unsigned)(x0 + kx) < (unsigned)g.W && pos >= 0 && ((both >> pos) & 1ull)) bits3 |= 1u << kx;
        }
        uint32_t pre = 0;
        if (bits3) pre = prefix[cs >> 5];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          int32_t val = -1;
          if ((bits3 >> kx) & 1u) {
            const int pos = sh + kx;
            const int32_t r = (int32_t)(pre + (uint32_t)__popcll(both & ((1ull << pos) - 1ull)));
            val = perm ? perm[r] - 1 : r;  // perm holds row + 1 (0 = empty)
          }
          v[(kz * 3 + ky) * 3 + kx] = val;
        }
      }
#pragma unroll
    for (int k = 0; k < 27; ++k) emit(v[k], k);
  } else if constexpr (KS != 0) {
    int32_t v[KS * KS * KS];
#pragma unroll
    for (int k = 0; k < KS * KS * KS; ++k) v[k] = lookup(k / (KS * KS), (k / KS) % KS, k % KS);
#pragma unroll
    for (int k = 0; k < KS * KS * KS; ++k) emit(v[k], k);
  } else {
    int k = 0;
    for (int kz = 0; kz < kd; ++kz)
      for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < kw; ++kx, ++k) emit(lookup(kz, ky, kx), k);
  }
  if (blockmask && (lane & 15) == 0 && o < n) blockmask[o >> 4] = mask;
  if (blk_cnt) {
    __syncthreads();
    const int kvol = kd * kh * kw;
    if ((int)threadIdx.x < kvol && threadIdx.x < 64)
      blk_cnt[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] =
          wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
  }
}

// one workgroup per offset: exclusive scan of the per-block counts (in place) and the pair count of
// the reference-format rulebook, indice_num[k] = total of offset kvol-1-k (geometry.h:247-297 order)
__global__ void __launch_bounds__(256)
block_offsets_kernel(uint32_t* __restrict__ blk, int64_t nblk, int kvol, int32_t* __restrict__ indice_num,
                     int flip = 1) {
  __shared__ uint32_t wsum[4];
  const int k = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (int64_t b0 = 0; b0 < nblk; b0 += 256) {
    const int64_t b = b0 + threadIdx.x;
    const uint32_t v = b < nblk ? blk[(int64_t)k * nblk + b] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) before += wsum[w];
      total += wsum[w];
    }
    if (b < nblk) blk[(int64_t)k * nblk + b] = carry + before + inc - v;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0 && indice_num) indice_num[flip ? kvol - 1 - k : k] = (int32_t)carry;
}

// pairs[k][0][pos] = j, pairs[k][1][pos] = nbr_t[K-1-k][j]  for valid entries
__global__ void __launch_bounds__(256)
compact_pairs_kernel(const int32_t* __restrict__ nbr_t, const uint32_t* __restrict__ blk_off,
                     int64_t n, int kvol, const int32_t* __restrict__ indice_num,
                     int32_t* __restrict__ pairs, int fill_tails, int flip = 1) {
  // grid (row blocks of 256, kvol): same blocking as neighbour_table_kernel, order preserving.
  // flip: the table is the gather side and the scatter side is its mirror image (plain sub-manifold);
  // !flip: the table already is the scatter side (row j -> its output at offset k)
  __shared__ uint32_t wsum[4];
  const int k = blockIdx.y;
  const int src = flip ? kvol - 1 - k : k;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // the unused tail of offset k reads -1 (ops.py:46-106 returns a -1 filled tensor); thread j owns slot j
  if (fill_tails && j < n && j >= indice_num[k]) {
    pairs[((int64_t)k * 2 + 0) * n + j] = -1;
    pairs[((int64_t)k * 2 + 1) * n + j] = -1;
  }
  const int32_t o = j < n ? nbr_t[(int64_t)src * n + j] : -1;
  const unsigned long long m = __ballot(o >= 0);
  if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (o < 0) return;
  uint32_t p = blk_off[(int64_t)src * gridDim.x + blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) p += wsum[w];
  pairs[((int64_t)k * 2 + 0) * n + p] = (int32_t)j;
  pairs[((int64_t)k * 2 + 1) * n + p] = o;
}

// generic rulebook -> gather table
__global__ void __launch_bounds__(256)
pairs_to_table_kernel(const int32_t* __restrict__ pairs, const int32_t* __restrict__ num,
                      int64_t cap, int side, int64_t rows, int32_t* __restrict__ table,
                      uint32_t* __restrict__ blockmask) {
  const int k = blockIdx.y;
  const int32_t nk = num[k];
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nk;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = pairs[((int64_t)k * 2 + 0) * cap + p];
    const int32_t b = pairs[((int64_t)k * 2 + 1) * cap + p];
    const int32_t dst = side ? b : a, val = side ? a : b;
    if ((unsigned)dst < (unsigned)rows) {
      table[(int64_t)k * rows + dst] = val;
      if (blockmask) atomicOr(blockmask + (dst >> 4), 1u << k);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Regular (strided) and transposed sparse conv rulebook: spconv::getIndicePair non-subM branch
// (spconv_ops.h:105-141; CPU functors geometry.h:144-245, GPU kernels indice.cu.h:22-145).
// Output rows are numbered in SORTED order of their flat grid index -- what the reference's
// GPU path produces through torch::_unique (spconv_ops.h:130); its CPU functor numbers them
// by first appearance instead.  Pairs inside an offset are in ascending input row, as on the
// CPU.  Same bitmap + popcount machinery as the sub-manifold case.
struct ConvGeom {
  int32_t batch, oD, oH, oW;
  int32_t kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int32_t transpose;
};

__device__ __forceinline__ bool conv_out_coord(int in, int c, int s, int p, int d, int osz, int transpose,
                                               int* out) {
  int o;
  if (transpose) {
    o = in * s - p + c * d;  // getValidOutPosTranspose, geometry.h:87-142
  } else {
    const int t = in + p - c * d;  // out * s - p + c * d = in, geometry.h:24-85
    if (t < 0 || t % s != 0) return false;
    o = t / s;
  }
  if (o < 0 || o >= osz) return false;
  *out = o;
  return true;
}

// grid (ceil(n/256), kvol): cand[k][j] = flat output cell reached from input j through offset k
__global__ void __launch_bounds__(256)
conv_mark_kernel(const int32_t* __restrict__ indices, int64_t n, ConvGeom g, int32_t* __restrict__ cand,
                 uint32_t* __restrict__ bitmap) {
  const int k = blockIdx.y;
  const int cx = k % g.kw, cy = (k / g.kw) % g.kh, cz = k / (g.kw * g.kh);
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int32_t b = indices[j * 4];
  int oz, oy, ox;
  int32_t cell = -1;
  if ((unsigned)b < (unsigned)g.batch &&
      conv_out_coord(indices[j * 4 + 1], cz, g.sd, g.pd, g.dd, g.oD, g.transpose, &oz) &&
      conv_out_coord(indices[j * 4 + 2], cy, g.sh, g.ph, g.dh, g.oH, g.transpose, &oy) &&
      conv_out_coord(indices[j * 4 + 3], cx, g.sw, g.pw, g.dw, g.oW, g.transpose, &ox)) {
    cell = (int32_t)((((int64_t)b * g.oD + oz) * g.oH + oy) * g.oW + ox);
    atomicOr(bitmap + (cell >> 5), 1u << (cell & 31));
  }
  cand[(int64_t)k * n + j] = cell;
}

__global__ void __launch_bounds__(256)
conv_emit_out_kernel(int64_t words, const uint32_t* __restrict__ bitmap,
                     const uint32_t* __restrict__ prefix, ConvGeom g, int32_t* __restrict__ out_indices,
                     int64_t capacity) {
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words;
       w += (int64_t)gridDim.x * blockDim.x) {
    uint32_t bits = bitmap[w];
    int64_t r = prefix[w];
    while (bits) {
      const int bb = __ffs(bits) - 1;
      bits &= bits - 1;
      if (r < capacity) {
        int64_t cell = w * 32 + bb;
        int32_t* o = out_indices + r * 4;
        o[3] = (int32_t)(cell % g.oW); cell /= g.oW;
        o[2] = (int32_t)(cell % g.oH); cell /= g.oH;
        o[1] = (int32_t)(cell % g.oD); cell /= g.oD;
        o[0] = (int32_t)cell;
      }
      ++r;
    }
  }
}

__global__ void __launch_bounds__(256)
conv_cell_to_rank_kernel(int64_t total, const uint32_t* __restrict__ bitmap,
                         const uint32_t* __restrict__ prefix, int32_t* __restrict__ cand) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t cell = cand[i];
    if (cell >= 0) cand[i] = rank_of(bitmap, prefix, cell);
  }
}

__global__ void __launch_bounds__(256)
conv_compact_pairs_kernel(const int32_t* __restrict__ cand, const uint32_t* __restrict__ pos, int64_t n,
                          int32_t* __restrict__ pairs) {
  const int k = blockIdx.y;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * blockDim.x) {
    const int32_t o = cand[(int64_t)k * n + j];
    if (o < 0) continue;
    const int64_t p = pos[(int64_t)k * n + j];
    pairs[((int64_t)k * 2 + 0) * n + p] = (int32_t)j;
    pairs[((int64_t)k * 2 + 1) * n + p] = o;
  }
}

__global__ void copy_totals_kernel(const uint32_t* __restrict__ totals, int kvol,
                                   int32_t* __restrict__ indice_num) {
  if ((int)threadIdx.x < kvol) indice_num[threadIdx.x] = (int32_t)totals[threadIdx.x];
}

struct ConvLayout {
  int64_t words, off_bitmap, off_prefix, off_cand, off_pos, off_scratch, off_totals, total;
};
inline bool make_conv_layout(int64_t n, int32_t batch, const int32_t* oshape, const int32_t* ksize,
                             ConvLayout* L) {
  if (batch < 1 || !oshape || !ksize || n < 0) return false;
  int64_t cells = batch;
  for (int i = 0; i < 3; ++i) {
    if (oshape[i] < 1 || ksize[i] < 1) return false;
    cells *= oshape[i];
    if (cells > 0x7fffffffLL) return false;
  }
  const int64_t kvol = (int64_t)ksize[0] * ksize[1] * ksize[2];
  L->words = (cells + 31) / 32;
  int64_t off = 0;
  auto take = [&](int64_t b) { int64_t o = off; off += ococc_align_up(b > 0 ? b : 4, 256); return o; };
  L->off_bitmap = take(L->words * 4);
  L->off_prefix = take(L->words * 4);
  L->off_cand = take(kvol * n * 4);
  L->off_pos = take(kvol * n * 4);
  const int64_t s1 = ococc_scan::scratch_words(L->words, 1);
  const int64_t s2 = ococc_scan::scratch_words(n > 0 ? n : 1, (int)kvol);
  L->off_scratch = take((s1 > s2 ? s1 : s2) * 4);
  L->off_totals = take((kvol + 1) * 4);
  L->total = off;
  return true;
}

struct Layout {
  int64_t words, off_bitmap, off_prefix, off_perm, off_pos, off_scratch, off_totals, total;
};

inline bool make_layout(int64_t n, int32_t batch, const int32_t* shape, const int32_t* ksize,
                        Layout* L) {
  if (batch < 1 || !shape || !ksize) return false;
  int64_t cells = batch;
  for (int i = 0; i < 3; ++i) {
    if (shape[i] < 1 || ksize[i] < 1 || !(ksize[i] & 1)) return false;
    cells *= shape[i];
    if (cells > 0x7fffffffLL) return false;
  }
  const int64_t kvol = (int64_t)ksize[0] * ksize[1] * ksize[2];
  L->words = (cells + 31) / 32;
  int64_t off = 0;
  L->off_bitmap = off;  off += ococc_align_up(L->words * 4, 256);
  L->off_perm = off;    off += ococc_align_up(n * 4, 256);  // directly behind the bitmap: one zero fill
  L->off_prefix = off;  off += ococc_align_up(L->words * 4, 256);
  L->off_pos = off;     off += ococc_align_up(kvol * ococc_cdiv(n > 0 ? n : 1, 256) * 4, 256);  // per-(offset, block) counts
  int64_t sw = ococc_scan::scratch_words(L->words, 1);
  L->off_scratch = off; off += ococc_align_up(sw * 4, 256);
  L->off_totals = off;  off += ococc_align_up((kvol + 1) * 4, 256);
  L->total = off;
  return true;
}

}  // namespace

extern "C" int64_t ococc_subm_rulebook_workspace_bytes(int64_t n, int32_t batch_size,
                                                       const int32_t host_shape[3],
                                                       const int32_t host_ksize[3]) {
  Layout L;
  if (n < 0 || !make_layout(n, batch_size, host_shape, host_ksize, &L)) return -1;
  return L.total;
}

namespace {

int subm_rulebook_impl(const int32_t* indices, int64_t n, int32_t batch_size, const int32_t host_shape[3],
                       const int32_t host_ksize[3], const int32_t host_dilation[3],
                       const uint32_t* grid_bitmap, const uint32_t* grid_prefix, int32_t* nbr_t,
                       uint32_t* blockmask, int32_t* indice_pairs, int32_t* indice_num, void* workspace,
                       int64_t workspace_bytes, hipStream_t stream, int fill_tails = 1) {
  Layout L;
  OCOCC_REQUIRE(n >= 0, "n < 0");
  OCOCC_REQUIRE(make_layout(n, batch_size, host_shape, host_ksize, &L),
                "need batch>=1, shape>=1, odd kernel sizes, batch*D*H*W < 2^31");
  int32_t dil[3] = {1, 1, 1};
  if (host_dilation)
    for (int i = 0; i < 3; ++i) {
      OCOCC_REQUIRE(host_dilation[i] >= 1, "dilation < 1");
      dil[i] = host_dilation[i];
    }
  const bool dilated = dil[0] != 1 || dil[1] != 1 || dil[2] != 1;
  OCOCC_REQUIRE(!dilated || !grid_bitmap, "the sorted-grid entry point is for dilation 1");
  const int kvol = host_ksize[0] * host_ksize[1] * host_ksize[2];
  OCOCC_REQUIRE(!blockmask || kvol <= 32, "blockmask needs kernel volume <= 32");
  if (n == 0) {
    if (indice_num) OCOCC_HIP(hipMemsetAsync(indice_num, 0, kvol * sizeof(int32_t), stream));
    return OCOCC_OK;
  }
  OCOCC_REQUIRE((indice_pairs == nullptr) == (indice_num == nullptr),
                "indice_pairs and indice_num go together");
  OCOCC_REQUIRE(indices && nbr_t, "null indices/nbr_t");
  OCOCC_REQUIRE(workspace && workspace_bytes >= L.total, "workspace too small");
  char* ws = (char*)workspace;
  const uint32_t* bitmap = grid_bitmap;
  const uint32_t* prefix = grid_prefix;
  const int32_t* perm = nullptr;  // identity when the caller's rows are already in cell order
  uint32_t* blk = (uint32_t*)(ws + L.off_pos);
  Geom g{batch_size, host_shape[0], host_shape[1], host_shape[2],
         host_ksize[0], host_ksize[1], host_ksize[2],
         -(host_ksize[0] / 2), -(host_ksize[1] / 2), -(host_ksize[2] / 2), dil[0], dil[1], dil[2]};
  const int64_t nblk = ococc_cdiv(n, 256);

  if (!grid_bitmap) {
    uint32_t* bm = (uint32_t*)(ws + L.off_bitmap);
    uint32_t* pf = (uint32_t*)(ws + L.off_prefix);
    int32_t* pm = (int32_t*)(ws + L.off_perm);
    uint32_t* scratch = (uint32_t*)(ws + L.off_scratch);
    // bitmap and perm (row + 1, 0 = empty) in one fill; blockmask and indice_num are fully written
    OCOCC_HIP(hipMemsetAsync(bm, 0, L.off_prefix - L.off_bitmap, stream));
    const int g1 = ococc_grid_1d(n, 256);
    hipLaunchKernelGGL(mark_voxels_kernel, dim3(g1), dim3(256), 0, stream, indices, n, g, bm,
                       (int32_t*)nullptr);
    OCOCC_CHECK_LAUNCH();
    OCOCC_HIP(ococc_scan::exclusive_scan<ococc_scan::POPC>(bm, L.words, L.words, 1, pf, L.words, scratch,
                                                           nullptr, stream));
    hipLaunchKernelGGL(fill_perm_kernel, dim3(g1), dim3(256), 0, stream, indices, n, g, bm, pf, pm);
    OCOCC_CHECK_LAUNCH();
    bitmap = bm;
    prefix = pf;
    perm = pm;
  }
  uint32_t* cnt = indice_pairs ? blk : nullptr;
  if (cnt && kvol > 64) OCOCC_HIP(hipMemsetAsync(blk, 0, (int64_t)kvol * nblk * 4, stream));
  if (dilated) {
    // The reference keeps padding = k/2 whatever the dilation, so the scatter side (input j -> outputs at
    // p + k/2 - m * dilation) is not the mirror image of the gather side.  Pass 1: scatter-side table into nbr_t,
    // compacted into the reference-format pairs (same ascending-j order as the CPU functor).  Pass 2: nbr_t is
    // overwritten with the gather-side table the convolution reads.
    if (indice_pairs) {
      Geom gs = g;
      gs.sz = host_ksize[0] / 2; gs.sy = host_ksize[1] / 2; gs.sx = host_ksize[2] / 2;
      gs.tz = -dil[0]; gs.ty = -dil[1]; gs.tx = -dil[2];
      hipLaunchKernelGGL(HIP_KERNEL_NAME(neighbour_table_kernel<0>), dim3((unsigned)nblk), dim3(256), 0, stream,
                         indices, n, gs, bitmap, prefix, perm, nbr_t, (uint32_t*)nullptr, cnt);
      OCOCC_CHECK_LAUNCH();
      hipLaunchKernelGGL(block_offsets_kernel, dim3(kvol), dim3(256), 0, stream, blk, nblk, kvol, indice_num, 0);
      OCOCC_CHECK_LAUNCH();
      hipLaunchKernelGGL(compact_pairs_kernel, dim3((unsigned)nblk, kvol), dim3(256), 0, stream, nbr_t, blk, n,
                         kvol, indice_num, indice_pairs, fill_tails, 0);
      OCOCC_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(neighbour_table_kernel<0>), dim3((unsigned)nblk), dim3(256), 0, stream,
                       indices, n, g, bitmap, prefix, perm, nbr_t, blockmask, (uint32_t*)nullptr);
    OCOCC_CHECK_LAUNCH();
    return OCOCC_OK;
  }
  if (g.kd == 3 && g.kh == 3 && g.kw == 3)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(neighbour_table_kernel<3>), dim3((unsigned)nblk), dim3(256), 0, stream,
                       indices, n, g, bitmap, prefix, perm, nbr_t, blockmask, cnt);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(neighbour_table_kernel<0>), dim3((unsigned)nblk), dim3(256), 0, stream,
                       indices, n, g, bitmap, prefix, perm, nbr_t, blockmask, cnt);
  OCOCC_CHECK_LAUNCH();
  if (indice_pairs) {
    hipLaunchKernelGGL(block_offsets_kernel, dim3(kvol), dim3(256), 0, stream, blk, nblk, kvol, indice_num);
    OCOCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(compact_pairs_kernel, dim3((unsigned)nblk, kvol), dim3(256), 0, stream, nbr_t, blk, n,
                       kvol, indice_num, indice_pairs, fill_tails);
    OCOCC_CHECK_LAUNCH();
  }
  return OCOCC_OK;
}

}  // namespace

extern "C" int ococc_subm_rulebook_build(const int32_t* indices, int64_t n, int32_t batch_size,
                                         const int32_t host_shape[3], const int32_t host_ksize[3],
                                         const int32_t host_dilation[3], int32_t* nbr_t,
                                         uint32_t* blockmask, int32_t* indice_pairs,
                                         int32_t* indice_num, void* workspace,
                                         int64_t workspace_bytes, ococc_stream_t stream) {
  return subm_rulebook_impl(indices, n, batch_size, host_shape, host_ksize, host_dilation, nullptr, nullptr,
                            nbr_t, blockmask, indice_pairs, indice_num, workspace, workspace_bytes,
                            (hipStream_t)stream);
}

extern "C" int ococc_subm_rulebook_build_sorted(const int32_t* indices, int64_t n, int32_t batch_size,
                                                const int32_t host_shape[3], const int32_t host_ksize[3],
                                                const uint32_t* grid_bitmap, const uint32_t* grid_prefix,
                                                int32_t* nbr_t, uint32_t* blockmask, int32_t* indice_pairs,
                                                int32_t* indice_num, int32_t fill_pair_tails, void* workspace,
                                                int64_t workspace_bytes, ococc_stream_t stream) {
  OCOCC_REQUIRE(grid_bitmap && grid_prefix, "null grid bitmap / prefix");
  return subm_rulebook_impl(indices, n, batch_size, host_shape, host_ksize, nullptr, grid_bitmap, grid_prefix,
                            nbr_t, blockmask, indice_pairs, indice_num, workspace, workspace_bytes,
                            (hipStream_t)stream, fill_pair_tails ? 1 : 0);
}

extern "C" int ococc_rulebook_pairs_to_table(const int32_t* indice_pairs,
                                             const int32_t* indice_num, int32_t kvol,
                                             int64_t pair_capacity, int32_t side,
                                             int64_t num_rows, int32_t* table,
                                             uint32_t* blockmask, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(kvol >= 1 && pair_capacity >= 0 && num_rows >= 0, "bad sizes");
  OCOCC_REQUIRE(side == 0 || side == 1, "side must be 0 or 1");
  OCOCC_REQUIRE(!blockmask || kvol <= 32, "blockmask needs kernel volume <= 32");
  if (num_rows == 0) return OCOCC_OK;
  OCOCC_REQUIRE(table, "null table");
  OCOCC_HIP(hipMemsetAsync(table, 0xff, (int64_t)kvol * num_rows * 4, stream));
  if (blockmask) OCOCC_HIP(hipMemsetAsync(blockmask, 0, ococc_cdiv(num_rows, 16) * 4, stream));
  if (pair_capacity == 0) return OCOCC_OK;
  OCOCC_REQUIRE(indice_pairs && indice_num, "null rulebook");
  hipLaunchKernelGGL(pairs_to_table_kernel, dim3(ococc_grid_1d(pair_capacity, 256, 1024), kvol),
                     dim3(256), 0, stream, indice_pairs, indice_num, pair_capacity, (int)side,
                     num_rows, table, blockmask);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int64_t ococc_conv_rulebook_workspace_bytes(int64_t n, int32_t batch_size,
                                                       const int32_t host_out_shape[3],
                                                       const int32_t host_ksize[3]) {
  ConvLayout L;
  if (!make_conv_layout(n, batch_size, host_out_shape, host_ksize, &L)) return -1;
  return L.total;
}

extern "C" int ococc_conv_rulebook_build(const int32_t* indices, int64_t n, int32_t batch_size,
                                         const int32_t host_out_shape[3], const int32_t host_ksize[3],
                                         const int32_t host_stride[3], const int32_t host_padding[3],
                                         const int32_t host_dilation[3], int32_t transpose,
                                         int32_t* out_indices, int64_t out_capacity,
                                         int32_t* indice_pairs, int32_t* indice_num, int32_t* num_out,
                                         void* workspace, int64_t workspace_bytes,
                                         ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvLayout L;
  OCOCC_REQUIRE(make_conv_layout(n, batch_size, host_out_shape, host_ksize, &L),
                "need batch>=1, out_shape>=1, batch*D*H*W < 2^31");
  OCOCC_REQUIRE(host_stride && host_padding && host_dilation, "null geometry");
  OCOCC_REQUIRE(indice_num && num_out, "null indice_num / num_out");
  const int kvol = host_ksize[0] * host_ksize[1] * host_ksize[2];
  for (int i = 0; i < 3; ++i)
    OCOCC_REQUIRE(host_stride[i] >= 1 && host_dilation[i] >= 1 && host_padding[i] >= 0, "bad geometry");
  OCOCC_HIP(hipMemsetAsync(indice_num, 0, kvol * sizeof(int32_t), stream));
  OCOCC_HIP(hipMemsetAsync(num_out, 0, sizeof(int32_t), stream));
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(indices && out_indices && indice_pairs, "null pointer");
  OCOCC_REQUIRE(workspace && workspace_bytes >= L.total, "workspace too small");
  char* ws = (char*)workspace;
  uint32_t* bitmap = (uint32_t*)(ws + L.off_bitmap);
  uint32_t* prefix = (uint32_t*)(ws + L.off_prefix);
  int32_t* cand = (int32_t*)(ws + L.off_cand);
  uint32_t* pos = (uint32_t*)(ws + L.off_pos);
  uint32_t* scratch = (uint32_t*)(ws + L.off_scratch);
  uint32_t* totals = (uint32_t*)(ws + L.off_totals);
  ConvGeom g{batch_size, host_out_shape[0], host_out_shape[1], host_out_shape[2],
             host_ksize[0], host_ksize[1], host_ksize[2], host_stride[0], host_stride[1], host_stride[2],
             host_padding[0], host_padding[1], host_padding[2], host_dilation[0], host_dilation[1],
             host_dilation[2], transpose ? 1 : 0};
  OCOCC_HIP(hipMemsetAsync(bitmap, 0, L.words * 4, stream));
  OCOCC_HIP(hipMemsetAsync(indice_pairs, 0xff, (int64_t)kvol * 2 * n * 4, stream));
  hipLaunchKernelGGL(conv_mark_kernel, dim3((unsigned)ococc_cdiv(n, 256), kvol), dim3(256), 0, stream,
                     indices, n, g, cand, bitmap);
  OCOCC_CHECK_LAUNCH();
  OCOCC_HIP(ococc_scan::exclusive_scan<ococc_scan::POPC>(bitmap, L.words, L.words, 1, prefix, L.words,
                                                         scratch, (uint32_t*)num_out, stream));
  hipLaunchKernelGGL(conv_emit_out_kernel, dim3(ococc_grid_1d(L.words, 256)), dim3(256), 0, stream,
                     L.words, bitmap, prefix, g, out_indices, out_capacity);
  OCOCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(conv_cell_to_rank_kernel, dim3(ococc_grid_1d((int64_t)kvol * n, 256)), dim3(256), 0,
                     stream, (int64_t)kvol * n, bitmap, prefix, cand);
  OCOCC_CHECK_LAUNCH();
  OCOCC_HIP(ococc_scan::exclusive_scan<ococc_scan::NONNEG>((const uint32_t*)cand, n, n, kvol, pos, n,
                                                           scratch, totals, stream));
  hipLaunchKernelGGL(conv_compact_pairs_kernel, dim3(ococc_grid_1d(n, 256, 1024), kvol), dim3(256), 0,
                     stream, cand, pos, n, indice_pairs);
  OCOCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(copy_totals_kernel, dim3(1), dim3(256), 0, stream, totals, kvol, indice_num);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
