/*
 * ococc_oracle.c -- CPU restatement of the reference algorithms on the OcOccNet
 * hot path (integer / index work and the fp32 sparse-conv loop).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or
 * executed by the product (objectcentricocccompletion_amd/); it is the checker
 * used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).  Pinning: the voxeliser functions are checked
 * against the reference's own C++ (compiled into oracle/_ref by
 * oracle/Makefile) through tests/golden/voxelize_*.npz; the rulebook functions
 * are pinned by the offset convention observed on the compiled reference
 * during the survey and by dense-convolution
 * equivalence (tests/test_oracle_cpu.py) -- the reference's rulebook code
 * itself cannot be built here (tensorview.h includes cuda_runtime_api.h, which
 * this image lacks), so for the rulebook "parity unpinned" applies beyond
 * those two checks.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ B1 ---- */
/* mmdet3d/ops/voxel/src/voxelization_cpu.cpp:8-41 (dynamic_voxelize_kernel)
 * and :145-169 (grid = ceil(range / voxel)).  coors are (z, y, x). */
void oracle_dynamic_voxelize(const float* points, int64_t n, int nf, const float* vs,
                             const float* range, int32_t* coors) {
  int grid[3];
  for (int j = 0; j < 3; ++j) grid[j] = (int)ceilf((range[3 + j] - range[j]) / vs[j]);
  for (int64_t i = 0; i < n; ++i) {
    for (int j = 0; j < 3; ++j) {
      int c = (int)floorf((points[i * nf + j] - range[j]) / vs[j]);
      if (c < 0) c = 0;
      else if (c >= grid[j]) c = grid[j] - 1;
      coors[i * 3 + (2 - j)] = c;
    }
  }
}

/* voxelization_cpu.cpp:44-99 (hard_voxelize_kernel) and :101-143
 * (grid = round(range / voxel)).  voxels [max_voxels,max_points,nf] and
 * num_points_per_voxel [max_voxels] must come in zeroed.  Returns voxel_num. */
int oracle_hard_voxelize(const float* points, int64_t n, int nf, const float* vs,
                         const float* range, int max_points, int max_voxels, float* voxels,
                         int32_t* coors, int32_t* num_points_per_voxel) {
  int grid[3];
  for (int j = 0; j < 3; ++j) grid[j] = (int)roundf((range[3 + j] - range[j]) / vs[j]);
  const int64_t cells = (int64_t)grid[0] * grid[1] * grid[2];
  int32_t* lut = (int32_t*)malloc(cells * sizeof(int32_t));
  for (int64_t c = 0; c < cells; ++c) lut[c] = -1;
  int voxel_num = 0;
  for (int64_t i = 0; i < n; ++i) {
    int c[3]; /* z, y, x */
    for (int j = 0; j < 3; ++j) {
      int v = (int)floorf((points[i * nf + j] - range[j]) / vs[j]);
      if (v < 0) v = 0;
      else if (v >= grid[j]) v = grid[j] - 1;
      c[2 - j] = v;
    }
    const int64_t cell = ((int64_t)c[0] * grid[1] + c[1]) * grid[0] + c[2];
    int vid = lut[cell];
    if (vid == -1) {
      vid = voxel_num;
      if (max_voxels != -1 && voxel_num >= max_voxels) continue;
      voxel_num += 1;
      lut[cell] = vid;
      for (int k = 0; k < 3; ++k) coors[vid * 3 + k] = c[k];
    }
    const int num = num_points_per_voxel[vid];
    if (max_points == -1 || num < max_points) {
      memcpy(voxels + ((int64_t)vid * max_points + num) * nf, points + i * nf, nf * sizeof(float));
      num_points_per_voxel[vid] += 1;
    }
  }
  free(lut);
  return voxel_num;
}

/* ------------------------------------------------------------------ B3 ---- */
/* include/spconv/geometry.h:24-85 (getValidOutPos): enumerate the output
 * positions an input position reaches, last dimension fastest, starting from
 * the upper corner; offset = sum_d m_d * (in_d - out_d*stride_d + pad_d)/dil_d
 * with m growing from the last dimension.  out: [kvol][4] = (z,y,x,offset). */
static int valid_out_pos(const int32_t* in, const int32_t* ks, const int32_t* stride,
                         const int32_t* pad, const int32_t* dil, const int32_t* oshape,
                         int32_t* out) {
  int32_t lo[3], up[3], cs[3], cnt[3] = {0, 0, 0};
  int npts = 1, np = 0;
  for (int d = 0; d < 3; ++d) {
    lo[d] = (in[d] - (ks[d] - 1) * dil[d] - 1 + stride[d] + pad[d]) / stride[d];
    up[d] = (in[d] + pad[d]) / stride[d];
    cs[d] = (up[d] - lo[d]) / dil[d] + 1;
    npts *= cs[d];
  }
  for (int i = 0; i < npts; ++i) {
    int valid = 1, m = 1, off = 0;
    for (int d = 2; d >= 0; --d) {
      const int32_t v = up[d] - cnt[d] * dil[d];
      out[np * 4 + d] = v;
      if (v < 0 || v > oshape[d] - 1) valid = 0;
      off += m * (in[d] - v * stride[d] + pad[d]) / dil[d];
      m *= ks[d];
    }
    out[np * 4 + 3] = off;
    if (valid) ++np;
    cnt[2] += 1;
    for (int d = 2; d > 0; --d)
      if (cnt[d] == cs[d]) { cnt[d - 1] += 1; cnt[d] = 0; }
  }
  return np;
}

/* geometry.h:247-297 (getIndicePairsSubM) with the sub-manifold parameters of
 * spconv_ops.h:77-82 (stride 1, pad = ksize/2).  indices [n,4] = (b,z,y,x);
 * pairs [kvol,2,n] pre-filled with -1 by the caller (spconv_ops.h:55-57),
 * num [kvol] zeroed.  Returns n. */
int64_t oracle_subm_rulebook(const int32_t* indices, int64_t n, int batch, const int32_t* shape,
                             const int32_t* ks, const int32_t* dil, int32_t* pairs,
                             int32_t* num) {
  const int64_t vol = (int64_t)shape[0] * shape[1] * shape[2];
  const int kvol = ks[0] * ks[1] * ks[2];
  int32_t* grid = (int32_t*)malloc(vol * batch * sizeof(int32_t));
  for (int64_t c = 0; c < vol * batch; ++c) grid[c] = -1;
  const int32_t stride[3] = {1, 1, 1};
  const int32_t pad[3] = {ks[0] / 2, ks[1] / 2, ks[2] / 2};
  int32_t* vp = (int32_t*)malloc((size_t)kvol * 4 * sizeof(int32_t));
  for (int64_t j = 0; j < n; ++j) {
    const int32_t* p = indices + j * 4;
    grid[(((int64_t)p[0] * shape[0] + p[1]) * shape[1] + p[2]) * shape[2] + p[3]] = (int32_t)j;
  }
  for (int64_t j = 0; j < n; ++j) {
    const int32_t* p = indices + j * 4;
    const int np = valid_out_pos(p + 1, ks, stride, pad, dil, shape, vp);
    for (int i = 0; i < np; ++i) {
      const int32_t* q = vp + i * 4;
      const int off = q[3];
      const int32_t o = grid[(((int64_t)p[0] * shape[0] + q[0]) * shape[1] + q[1]) * shape[2] + q[2]];
      if (o > -1) {
        pairs[((int64_t)off * 2 + 0) * n + num[off]] = (int32_t)j;
        pairs[((int64_t)off * 2 + 1) * n + num[off]] = o;
        num[off] += 1;
      }
    }
  }
  free(vp);
  free(grid);
  return n;
}

/* geometry.h:144-193 (getIndicePairsConv): regular (strided) sparse conv on
 * the CPU: outputs are numbered in order of first appearance.  out_indices
 * [n*kvol,4], pairs [kvol,2,n] (-1 filled), num [kvol] zeroed.  Returns the
 * number of active outputs. */
int64_t oracle_conv_rulebook(const int32_t* indices, int64_t n, int batch, const int32_t* oshape,
                             const int32_t* ks, const int32_t* stride, const int32_t* pad,
                             const int32_t* dil, int32_t* out_indices, int32_t* pairs,
                             int32_t* num) {
  const int64_t vol = (int64_t)oshape[0] * oshape[1] * oshape[2];
  const int kvol = ks[0] * ks[1] * ks[2];
  int32_t* grid = (int32_t*)malloc(vol * batch * sizeof(int32_t));
  for (int64_t c = 0; c < vol * batch; ++c) grid[c] = -1;
  int32_t* vp = (int32_t*)malloc((size_t)kvol * 4 * sizeof(int32_t));
  int64_t nact = 0;
  for (int64_t j = 0; j < n; ++j) {
    const int32_t* p = indices + j * 4;
    const int np = valid_out_pos(p + 1, ks, stride, pad, dil, oshape, vp);
    for (int i = 0; i < np; ++i) {
      const int32_t* q = vp + i * 4;
      const int off = q[3];
      const int64_t cell = (((int64_t)p[0] * oshape[0] + q[0]) * oshape[1] + q[1]) * oshape[2] + q[2];
      if (grid[cell] == -1) {
        out_indices[nact * 4 + 0] = p[0];
        for (int k = 0; k < 3; ++k) out_indices[nact * 4 + 1 + k] = q[k];
        grid[cell] = (int32_t)nact++;
      }
      pairs[((int64_t)off * 2 + 0) * n + num[off]] = (int32_t)j;
      pairs[((int64_t)off * 2 + 1) * n + num[off]] = grid[cell];
      num[off] += 1;
    }
  }
  free(vp);
  free(grid);
  return nact;
}

/* geometry.h:87-142 (getValidOutPosTranspose) + :195-245 (getIndicePairsDeConv): transposed
 * sparse conv, outputs numbered by first appearance. */
int64_t oracle_deconv_rulebook(const int32_t* indices, int64_t n, int batch, const int32_t* oshape,
                               const int32_t* ks, const int32_t* stride, const int32_t* pad,
                               const int32_t* dil, int32_t* out_indices, int32_t* pairs, int32_t* num) {
  const int64_t vol = (int64_t)oshape[0] * oshape[1] * oshape[2];
  int32_t* grid = (int32_t*)malloc(vol * batch * sizeof(int32_t));
  for (int64_t c = 0; c < vol * batch; ++c) grid[c] = -1;
  int64_t nact = 0;
  for (int64_t j = 0; j < n; ++j) {
    const int32_t* p = indices + j * 4;
    int32_t lo[3], cs[3], cnt[3] = {0, 0, 0};
    int npts = 1;
    for (int d = 0; d < 3; ++d) {
      lo[d] = p[1 + d] * stride[d] - pad[d];
      cs[d] = ks[d];  /* (uppers - lowers) / dil + 1 with uppers = lowers + (k-1)*dil */
      npts *= cs[d];
    }
    for (int i = 0; i < npts; ++i) {
      int valid = 1, m = 1, off = 0;
      int32_t q[3];
      for (int d = 2; d >= 0; --d) {
        const int32_t up = lo[d] + (ks[d] - 1) * dil[d];
        const int32_t v = up - cnt[d] * dil[d];
        q[d] = v;
        if (v < 0 || v > oshape[d] - 1) valid = 0;
        off += m * (v - lo[d]) / dil[d];
        m *= ks[d];
      }
      if (valid) {
        const int64_t cell = (((int64_t)p[0] * oshape[0] + q[0]) * oshape[1] + q[1]) * oshape[2] + q[2];
        if (grid[cell] == -1) {
          out_indices[nact * 4] = p[0];
          for (int k = 0; k < 3; ++k) out_indices[nact * 4 + 1 + k] = q[k];
          grid[cell] = (int32_t)nact++;
        }
        pairs[((int64_t)off * 2 + 0) * n + num[off]] = (int32_t)j;
        pairs[((int64_t)off * 2 + 1) * n + num[off]] = grid[cell];
        num[off] += 1;
      }
      cnt[2] += 1;
      for (int d = 2; d > 0; --d)
        if (cnt[d] == cs[d]) { cnt[d - 1] += 1; cnt[d] = 0; }
    }
  }
  free(grid);
  return nact;
}

/* ------------------------------------------------------------------ B4 ---- */
static int argmax_first(const int32_t* num, int kvol) {
  int best = 0;
  for (int k = 1; k < kvol; ++k)
    if (num[k] > num[best]) best = k;
  return best;
}

/* C[m,n] (+)= A[m,k] * B[k,n], row major, double accumulation per element */
static void mm(const float* A, const float* B, float* C, int64_t m, int k, int n, int accumulate) {
  for (int64_t i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int t = 0; t < k; ++t) s += (double)A[i * k + t] * (double)B[(int64_t)t * n + j];
      C[i * n + j] = accumulate ? C[i * n + j] + (float)s : (float)s;
    }
}

/* include/spconv/spconv_ops.h:260-361 (indiceConv<T>, CPU branch) with the
 * CPU gather / scatter-add functors of src/reordering.cc:21-50.
 * features [n_in,cin], filters [kvol,cin,cout], pairs [kvol,2,cap],
 * out [n_out,cout] (overwritten).  inverse swaps the roles of the two pair
 * rows; subm sends the fullest offset (the centre) through one dense mm. */
void oracle_indice_conv(const float* features, int64_t n_in, int cin, const float* filters,
                        int kvol, int cout, const int32_t* pairs, const int32_t* num, int64_t cap,
                        int64_t n_out, int inverse, int subm, float* out) {
  memset(out, 0, (size_t)n_out * cout * sizeof(float));
  const int centre = argmax_first(num, kvol);
  if (subm) mm(features, filters + (int64_t)centre * cin * cout, out, n_out, cin, cout, 0);
  int32_t maxn = 0;
  for (int k = 0; k < kvol; ++k) if (num[k] > maxn) maxn = num[k];
  float* ibuf = (float*)malloc((size_t)(maxn > 0 ? maxn : 1) * cin * sizeof(float));
  float* obuf = (float*)malloc((size_t)(maxn > 0 ? maxn : 1) * cout * sizeof(float));
  for (int k = 0; k < kvol; ++k) {
    const int nh = num[k];
    if (nh <= 0 || (subm && k == centre)) continue;
    const int32_t* gi = pairs + ((int64_t)k * 2 + (inverse ? 1 : 0)) * cap;
    const int32_t* so = pairs + ((int64_t)k * 2 + (inverse ? 0 : 1)) * cap;
    for (int p = 0; p < nh; ++p)
      memcpy(ibuf + (int64_t)p * cin, features + (int64_t)gi[p] * cin, cin * sizeof(float));
    mm(ibuf, filters + (int64_t)k * cin * cout, obuf, nh, cin, cout, 0);
    for (int p = 0; p < nh; ++p)
      for (int c = 0; c < cout; ++c) out[(int64_t)so[p] * cout + c] += obuf[(int64_t)p * cout + c];
  }
  free(ibuf);
  free(obuf);
}

/* spconv_ops.h:363-456 (indiceConvBackward<T>, CPU branch).
 * din [n_in,cin] and dfilters [kvol,cin,cout] are overwritten. */
void oracle_indice_conv_backward(const float* features, int64_t n_in, int cin,
                                 const float* filters, int kvol, int cout, const float* dout,
                                 int64_t n_out, const int32_t* pairs, const int32_t* num,
                                 int64_t cap, int inverse, int subm, float* din, float* dfilters) {
  memset(din, 0, (size_t)n_in * cin * sizeof(float));
  memset(dfilters, 0, (size_t)kvol * cin * cout * sizeof(float));
  const int centre = argmax_first(num, kvol);
  if (subm) {
    /* dW[c] = X^T dY ; dX = dY W[c]^T */
    float* dw = dfilters + (int64_t)centre * cin * cout;
    const float* w = filters + (int64_t)centre * cin * cout;
    for (int a = 0; a < cin; ++a)
      for (int b = 0; b < cout; ++b) {
        double s = 0.0;
        for (int64_t r = 0; r < n_out; ++r) s += (double)features[r * cin + a] * dout[r * cout + b];
        dw[a * cout + b] = (float)s;
      }
    for (int64_t r = 0; r < n_out; ++r)
      for (int a = 0; a < cin; ++a) {
        double s = 0.0;
        for (int b = 0; b < cout; ++b) s += (double)dout[r * cout + b] * w[a * cout + b];
        din[r * cin + a] = (float)s;
      }
  }
  for (int k = 0; k < kvol; ++k) {
    const int nh = num[k];
    if (nh <= 0 || (subm && k == centre)) continue;
    const int32_t* gi = pairs + ((int64_t)k * 2 + (inverse ? 1 : 0)) * cap;
    const int32_t* go = pairs + ((int64_t)k * 2 + (inverse ? 0 : 1)) * cap;
    float* dw = dfilters + (int64_t)k * cin * cout;
    const float* w = filters + (int64_t)k * cin * cout;
    for (int a = 0; a < cin; ++a)
      for (int b = 0; b < cout; ++b) {
        double s = 0.0;
        for (int p = 0; p < nh; ++p)
          s += (double)features[(int64_t)gi[p] * cin + a] * dout[(int64_t)go[p] * cout + b];
        dw[a * cout + b] = (float)s;
      }
    for (int p = 0; p < nh; ++p)
      for (int a = 0; a < cin; ++a) {
        double s = 0.0;
        for (int b = 0; b < cout; ++b) s += (double)dout[(int64_t)go[p] * cout + b] * w[a * cout + b];
        din[(int64_t)gi[p] * cin + a] += (float)s;
      }
  }
}

/* ------------------------------------------------------------- A5 / B2 ---- */
/* Segment reduce given a dense inverse map: the arithmetic of
 * feats_reduce_kernel (mmdet3d/ops/voxel/src/scatter_points_cuda.cu:81-103)
 * and of torch_scatter.scatter_max / scatter(reduce=mean|sum) as called at
 * mmdet3d/ops/sst/sst_ops.py:171-174.  reduce: 0 sum, 1 mean, 2 max.
 * arg (max only, may be NULL): smallest row index attaining the max
 * (scatter_points_cuda.cu:136-160).  Rows with inv < 0 are ignored. */
void oracle_segment_reduce(const float* feats, const int32_t* inv, int64_t n, int c, int reduce,
                           int64_t segs, float* out, int32_t* counts, int32_t* arg) {
  for (int64_t s = 0; s < segs; ++s) counts[s] = 0;
  for (int64_t i = 0; i < segs * c; ++i) out[i] = reduce == 2 ? -INFINITY : 0.f;
  if (arg) for (int64_t i = 0; i < segs * c; ++i) arg[i] = 0x7f7f7f7f;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t s = inv[i];
    if (s < 0) continue;
    counts[s] += 1;
    for (int ch = 0; ch < c; ++ch) {
      const float v = feats[i * c + ch];
      float* o = out + (int64_t)s * c + ch;
      if (reduce == 2) {
        if (v > *o) { *o = v; if (arg) arg[(int64_t)s * c + ch] = (int32_t)i; }
      } else {
        *o += v;
      }
    }
  }
  for (int64_t s = 0; s < segs; ++s)
    for (int ch = 0; ch < c; ++ch) {
      float* o = out + s * c + ch;
      if (reduce == 1) *o = counts[s] > 0 ? *o / (float)counts[s] : 0.f;
      if (reduce == 2 && counts[s] == 0) *o = 0.f;
    }
}

/* Sorted unique rows with inverse and counts: what at::unique_dim(coors, 0,
 * sorted=true, return_inverse, return_counts) gives
 * (scatter_points_cuda.cu:199-210), with rows containing a negative entry
 * dropped (:202,208-210).  Returns the number of unique rows. */
typedef struct { const int32_t* base; int ndim; } row_ctx;
static row_ctx g_ctx;
static int cmp_rows(const void* a, const void* b) {
  const int32_t* ra = g_ctx.base + (int64_t)(*(const int64_t*)a) * g_ctx.ndim;
  const int32_t* rb = g_ctx.base + (int64_t)(*(const int64_t*)b) * g_ctx.ndim;
  for (int k = 0; k < g_ctx.ndim; ++k) {
    if (ra[k] < rb[k]) return -1;
    if (ra[k] > rb[k]) return 1;
  }
  return 0;
}
int64_t oracle_unique_rows(const int32_t* coors, int64_t n, int ndim, int32_t* out_coors,
                           int32_t* inv, int32_t* counts) {
  int64_t* order = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    int neg = 0;
    for (int k = 0; k < ndim; ++k) neg |= coors[i * ndim + k] < 0;
    if (neg) inv[i] = -1; else order[m++] = i;
  }
  g_ctx.base = coors;
  g_ctx.ndim = ndim;
  qsort(order, (size_t)m, sizeof(int64_t), cmp_rows);
  int64_t u = 0;
  for (int64_t t = 0; t < m; ++t) {
    const int64_t i = order[t];
    if (t == 0 || cmp_rows(&order[t - 1], &order[t]) != 0) {
      memcpy(out_coors + u * ndim, coors + i * ndim, ndim * sizeof(int32_t));
      counts[u] = 0;
      ++u;
    }
    inv[i] = (int32_t)(u - 1);
    counts[u - 1] += 1;
  }
  free(order);
  return u;
}

/* ------------------------------------------------------------------ A3 ---- */
/* Points-in-rotated-box pooling with the semantics documented in
 * objectcentricocccompletion_amd/csrc/point_pool.hip (TorchEx source absent: the contract
 * comes from mmdet3d/ops/dynamic_point_pool_op.py:63-113 and the assertions of
 * dynamic_point_roi_extractor.py:222-234; geometry as mmdet3d's check_pt_in_box3d).
 * Rows sorted by (roi, point).  Returns the number of rows written. */
int64_t oracle_point_pool(const float* rois, const int32_t* roi_key, int64_t R, const float* pts,
                          const int32_t* pts_key, int64_t N, const float* extra, int max_inbox,
                          int64_t max_all, int64_t* out_pts, int64_t* out_roi, float* feats,
                          int32_t* roi_counts) {
  int64_t m = 0;
  for (int64_t r = 0; r < R; ++r) {
    const float* b = rois + r * 7;
    const float w = b[3], l = b[4], h = b[5], rz = b[6];
    const float cx = b[0], cy = b[1], cz = b[2] + h * 0.5f;
    const float hw = w * 0.5f, hl = l * 0.5f, hh = h * 0.5f;
    const float ehw = (w + extra[0]) * 0.5f, ehl = (l + extra[1]) * 0.5f, ehh = (h + extra[2]) * 0.5f;
    const float cosa = cosf(-rz), sina = sinf(-rz);
    int kept = 0;
    for (int64_t i = 0; i < N && kept < max_inbox && m < max_all; ++i) {
      if (pts_key[i] != roi_key[r]) continue;
      const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
      const float dz = z - cz;
      if (fabsf(dz) > ehh) continue;
      const float sx = x - cx, sy = y - cy;
      const float px = sx * cosa + sy * (-sina);
      const float py = sx * sina + sy * cosa;
      if (!(px > -ehl && px < ehl && py > -ehw && py < ehw)) continue;
      const int inner = fabsf(dz) <= hh && px > -hl && px < hl && py > -hw && py < hw;
      float* f = feats + m * 13;
      f[0] = x; f[1] = y; f[2] = z; f[3] = px; f[4] = py; f[5] = dz;
      f[6] = px + hl; f[7] = py + hw; f[8] = dz + hh; f[9] = hl - px; f[10] = hw - py; f[11] = hh - dz;
      f[12] = inner ? 0.f : 1.f;
      out_pts[m] = i;
      out_roi[m] = r;
      ++m;
      ++kept;
    }
    if (roi_counts) roi_counts[r] = kept;
  }
  return m;
}

/* ------------------------------------------------------------------ A2 ---- */
/* aligned_iou_3d (mmdet3d/core/bbox/structures/lidar_box3d.py:404-448): BEV overlap of two
 * rotated rectangles (w along x at yaw 0, clockwise yaw -- the iou3d convention, the TorchEx
 * source being absent) times the height overlap, over the union volume (clamp 1e-8).
 * Independent formulation from the HIP kernel: the intersection polygon is collected as the
 * vertices of either rectangle inside the other plus all edge-edge intersection points,
 * sorted by angle around their centroid. */
static void box_corners(const float* b, double* cx, double* cy) {
  const double hw = b[3] * 0.5, hl = b[4] * 0.5, ca = cos((double)b[6]), sa = sin((double)b[6]);
  const double dx[4] = {-hw, hw, hw, -hw}, dy[4] = {-hl, -hl, hl, hl};
  for (int k = 0; k < 4; ++k) {
    cx[k] = dx[k] * ca + dy[k] * sa + b[0];
    cy[k] = -dx[k] * sa + dy[k] * ca + b[1];
  }
}
static int inside_quad(const double* qx, const double* qy, double px, double py) {
  int pos = 0, neg = 0;
  for (int i = 0; i < 4; ++i) {
    const int j = (i + 1) & 3;
    const double c = (qx[j] - qx[i]) * (py - qy[i]) - (qy[j] - qy[i]) * (px - qx[i]);
    if (c > 1e-12) pos = 1;
    if (c < -1e-12) neg = 1;
  }
  return !(pos && neg);
}
/* area of the intersection of two rotated BEV rectangles given as (x, y, z, w, l, h, yaw) rows */
static double bev_overlap_area(const float* a, const float* b) {
  double ax[4], ay[4], bx[4], by[4], px[24], py[24];
  int m = 0;
  box_corners(a, ax, ay);
  box_corners(b, bx, by);
  for (int i = 0; i < 4; ++i) if (inside_quad(bx, by, ax[i], ay[i])) { px[m] = ax[i]; py[m++] = ay[i]; }
  for (int i = 0; i < 4; ++i) if (inside_quad(ax, ay, bx[i], by[i])) { px[m] = bx[i]; py[m++] = by[i]; }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const int i2 = (i + 1) & 3, j2 = (j + 1) & 3;
      const double rx = ax[i2] - ax[i], ry = ay[i2] - ay[i], sx = bx[j2] - bx[j], sy = by[j2] - by[j];
      const double den = rx * sy - ry * sx;
      if (fabs(den) < 1e-14) continue;
      const double u = ((bx[j] - ax[i]) * sy - (by[j] - ay[i]) * sx) / den;
      const double v = ((bx[j] - ax[i]) * ry - (by[j] - ay[i]) * rx) / den;
      if (u >= 0 && u <= 1 && v >= 0 && v <= 1) { px[m] = ax[i] + u * rx; py[m++] = ay[i] + u * ry; }
    }
  double area = 0.0;
  if (m >= 3) {
    double gx = 0, gy = 0;
    for (int i = 0; i < m; ++i) { gx += px[i]; gy += py[i]; }
    gx /= m; gy /= m;
    for (int i = 1; i < m; ++i)  /* insertion sort by angle */
      for (int j = i; j > 0 && atan2(py[j] - gy, px[j] - gx) < atan2(py[j - 1] - gy, px[j - 1] - gx); --j) {
        double tx = px[j], ty = py[j]; px[j] = px[j - 1]; py[j] = py[j - 1]; px[j - 1] = tx; py[j - 1] = ty;
      }
    for (int i = 0; i < m; ++i) { const int j = (i + 1) % m; area += px[i] * py[j] - px[j] * py[i]; }
    area = fabs(area) * 0.5;
  }
  return area;
}

void oracle_aligned_iou3d(const float* b1, const float* b2, int64_t n, float* iou) {
  for (int64_t t = 0; t < n; ++t) {
    const float* a = b1 + t * 7;
    const float* b = b2 + t * 7;
    const double area = bev_overlap_area(a, b);
    double top = fmin((double)a[2] + a[5], (double)b[2] + b[5]), bot = fmax((double)a[2], (double)b[2]);
    double oh = top - bot; if (oh < 0) oh = 0;
    const double inter = area * oh, v1 = (double)a[3] * a[4] * a[5], v2 = (double)b[3] * b[4] * b[5];
    double den = v1 + v2 - inter; if (den < 1e-8) den = 1e-8;
    iou[t] = (float)(inter / den);
  }
}

/* TorchEx boxes_overlap_1to1 contract (call site lidar_box3d.py:429-434): rows (x1, y1, x2, y2, yaw) as
 * xywhr2xyxyr makes them (structures/utils.py:85-103) -> BEV intersection area of the i-th pair. */
void oracle_bev_overlap_1to1(const float* r1, const float* r2, int64_t n, float* area) {
  for (int64_t t = 0; t < n; ++t) {
    const float* p = r1 + t * 5;
    const float* q = r2 + t * 5;
    const float a[7] = {(p[0] + p[2]) * 0.5f, (p[1] + p[3]) * 0.5f, 0.f, p[2] - p[0], p[3] - p[1], 1.f, p[4]};
    const float b[7] = {(q[0] + q[2]) * 0.5f, (q[1] + q[3]) * 0.5f, 0.f, q[2] - q[0], q[3] - q[1], 1.f, q[4]};
    area[t] = (float)bev_overlap_area(a, b);
  }
}
