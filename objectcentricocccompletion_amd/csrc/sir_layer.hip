// A whole SIRLayer (mmdet3d/models/voxel_encoders/voxel_encoder.py:764-832) behind ONE call per direction: the
// sequence of point_mlp launches that objectcentricocccompletion_amd/sir.py issued from Python -- rel_mlp blocks, vfe
// blocks with their segment maxima, the concatenation of the maxima, the shortcut; and in reverse the backward launches,
// the arg-max passes, the weight-gradient slices and the few element-wise joins between them.  Nothing new is computed
// here: at 4 tracklets the training step is bound by the host, and a layer cost ~0.27 ms (forward) + ~0.5 ms (backward)
// of interpreter time around ~25 launches.  Intermediates live in one caller-owned slab per direction.
#include "common.hpp"
#include "sir_fused.hpp"

namespace {

constexpr int kMaxBlocks = 8;

inline int64_t pad64(int64_t v) { return ococc_align_up(v, 64); }

struct Dims {
  int nl, nr, nv;
  int n[kMaxBlocks], k[kMaxBlocks];
  int sum_n;   // columns of the concatenated maxima
};

inline int read_dims(const ococc_sir_layer* d, Dims* D) {
  OCOCC_REQUIRE(d, "null descriptor");
  OCOCC_REQUIRE(d->n_rel >= 0 && d->n_vfe >= 1 && d->n_rel + d->n_vfe <= kMaxBlocks, "0 <= rel blocks, 1 <= vfe blocks, <= 8 in all");
  OCOCC_REQUIRE(d->feat_cols >= 1 && d->cluster_cols >= 0, "bad column counts");
  D->nr = d->n_rel;
  D->nv = d->n_vfe;
  D->nl = D->nr + D->nv;
  D->sum_n = 0;
  for (int b = 0; b < D->nl; ++b) {
    OCOCC_REQUIRE(d->n[b] >= 1, "bad block width");
    D->n[b] = d->n[b];
    if (b < D->nr)
      D->k[b] = b == 0 ? d->cluster_cols : d->n[b - 1];
    else if (b == D->nr)
      D->k[b] = d->feat_cols + (d->with_cluster_center ? d->cluster_cols : 0);
    else
      D->k[b] = 2 * d->n[b - 1];
    if (b >= D->nr) D->sum_n += d->n[b];
  }
  OCOCC_REQUIRE(D->nr == 0 || d->n[D->nr - 1] == d->feat_cols, "the gate is as wide as the features");
  OCOCC_REQUIRE(!d->shortcut || d->n[D->nl - 1] == d->feat_cols - 3, "the shortcut adds the non-xyz feature columns");
  return OCOCC_OK;
}

struct FwdLayout {
  int64_t y[kMaxBlocks], m[kMaxBlocks], arg[kMaxBlocks], total;
};
inline void fwd_layout(const Dims& D, int64_t rows, int64_t groups, FwdLayout* L) {
  int64_t off = 0;
  for (int b = 0; b < D.nl; ++b) {
    L->y[b] = off;
    off += pad64(rows * D.n[b]);
  }
  for (int i = 0; i < D.nv; ++i) {
    L->m[i] = off;
    off += pad64(groups * D.n[D.nr + i]);
  }
  for (int i = 0; i < D.nv; ++i) {   // (int32) the smallest row attaining each maximum: written by the forward pass
    L->arg[i] = off;
    off += pad64(groups * D.n[D.nr + i]);
  }
  L->total = off > 0 ? off : 64;
}

struct BwdLayout {
  int64_t dz[kMaxBlocks], xcat[kMaxBlocks], da[2], dab[kMaxBlocks], dgate, dv[kMaxBlocks], dm, dy0, lnp[kMaxBlocks], wp[kMaxBlocks], total;
  int64_t tiles;
  int slices;
};
inline void bwd_layout(const Dims& D, int feat_cols, int64_t rows, int64_t groups, BwdLayout* L) {
  int max_n = 0, max_ka = 1;
  for (int b = 0; b < D.nl; ++b) {
    max_n = D.n[b] > max_n ? D.n[b] : max_n;
    if (b != 0 && b != D.nr) max_ka = D.n[b - 1] > max_ka ? D.n[b - 1] : max_ka;   // blocks fed by another block's y
  }
  L->tiles = ococc_point_mlp_tiles(rows);
  L->slices = ococc_point_mlp_wgrad_slices(rows);
  int64_t off = 0;
  auto take = [&](int64_t c) { int64_t o = off; off += pad64(c > 0 ? c : 1); return o; };
  // (dz and the assembled input of EVERY block stay until the end of the pass: the blocks' weight-gradient products
  // then share one launch)
  for (int b = 0; b < D.nl; ++b) L->dz[b] = take(rows * D.n[b]);
  for (int b = 0; b < D.nl; ++b) L->xcat[b] = take(rows * D.k[b]);
  L->da[0] = take(rows * max_ka);
  L->da[1] = take(rows * max_ka);
  // (one launch per layer: every block fed by another block has the gradient of its input rows in a buffer of its own)
  for (int b = 0; b < D.nl; ++b) L->dab[b] = (b != 0 && b != D.nr) ? take(rows * D.n[b - 1]) : 0;
  L->dgate = take(rows * feat_cols);
  for (int i = 0; i < D.nv; ++i) L->dv[i] = i > 0 ? take(groups * D.n[D.nr + i - 1]) : 0;
  L->dm = take(groups * max_n);
  L->dy0 = take(rows * D.n[D.nl - 1]);
  for (int b = 0; b < D.nl; ++b) L->lnp[b] = take(L->tiles * 2 * D.n[b]);
  for (int b = 0; b < D.nl; ++b) L->wp[b] = take((int64_t)L->slices * D.n[b] * D.k[b]);
  L->total = off;
}

// out[r][c] = y[r][c] + feats[r][3 + c]
__global__ void __launch_bounds__(256)
shortcut_kernel(const float* __restrict__ y, const float* __restrict__ feats, int lda, int n, int64_t count,
                float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / n;
    const int c = (int)(i - r * n);
    out[i] = y[i] + feats[r * lda + 3 + c];
  }
}
// dfeat[r][3 + c] += dy[r][c]
__global__ void __launch_bounds__(256)
shortcut_grad_kernel(float* __restrict__ dfeat, int ldf, const float* __restrict__ dy, int n, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / n;
    const int c = (int)(i - r * n);
    dfeat[r * ldf + 3 + c] += dy[i];
  }
}
// out[g][off + c] = m[g][c]: one source of the concatenated maxima
__global__ void __launch_bounds__(256)
place_cols_kernel(const float* __restrict__ m, int n, int64_t count, float* __restrict__ out, int ld, int off) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const int64_t g = i / n;
    const int c = (int)(i - g * n);
    out[g * ld + off + c] = m[i];
  }
}
// out[g][c] = (a ? a[g * lda + c] : 0) + (b ? b[g][c] : 0): the gradient of a block's maxima = the slice of the incoming
// one + what the next block's gathered copy received
__global__ void __launch_bounds__(256)
join_cols_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int n, int64_t count,
                 float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const int64_t g = i / n;
    const int c = (int)(i - g * n);
    out[i] = (a ? a[g * lda + c] : 0.f) + (b ? b[i] : 0.f);
  }
}

// the one-launch form's view of a layer: the descriptor + where the forward slab keeps every block's rows
void fused_args(const ococc_sir_layer* d, const Dims& D, const FwdLayout& F, const float* feats, const float* f_cluster,
                const int32_t* inv, int64_t rows, int64_t groups, float* fwd_slab, float* y_out, SirFusedArgs* A) {
  *A = SirFusedArgs{};
  A->feats = feats;
  A->fc = f_cluster;
  A->inv = inv;
  A->rel_cs = d->rel_colscale;
  A->col = d->colscale;
  A->gate = D.nr == 0 ? d->gate : nullptr;
  A->ld_gate = d->feat_cols;
  A->feat_cols = d->feat_cols;
  A->cluster_cols = d->cluster_cols;
  A->with_cc = d->with_cluster_center;
  A->shortcut = d->shortcut;
  A->nr = D.nr;
  A->nv = D.nv;
  A->sum_n = D.sum_n;
  A->bscale = d->bscale;
  A->rows = rows;
  A->groups = groups;
  const int last = D.nl - 1;
  for (int b = 0; b < D.nl; ++b) {
    SirBlockArgs& B = A->b[b];
    B.wf = d->w_frag[b];
    B.wtf = d->wt_frag[b];
    B.ln_w = d->ln_weight[b];
    B.ln_b = d->ln_bias[b];
    B.eps = d->eps[b];
    B.n = D.n[b];
    B.k = D.k[b];
    B.act = d->act[b];
    B.y = (b == last && !d->shortcut) ? y_out : fwd_slab + F.y[b];
    if (b >= D.nr) {
      B.m = fwd_slab + F.m[b - D.nr];
      B.arg = (int32_t*)(fwd_slab + F.arg[b - D.nr]);
    }
  }
  A->y_out = y_out;
}

}  // namespace

extern "C" int64_t ococc_sir_layer_fwd_floats(const ococc_sir_layer* d, int64_t rows, int64_t groups) {
  Dims D;
  if (read_dims(d, &D) != OCOCC_OK || rows < 0 || groups < 0) return -1;
  FwdLayout L;
  fwd_layout(D, rows, groups, &L);
  return L.total;
}

extern "C" int ococc_sir_layer_bwd_layout(const ococc_sir_layer* d, int64_t rows, int64_t groups, int64_t* ln_partial_off,
                                          int64_t* w_partial_off, int64_t* tiles, int32_t* slices, int64_t* total) {
  Dims D;
  if (int rc = read_dims(d, &D)) return rc;
  OCOCC_REQUIRE(rows >= 0 && groups >= 0 && ln_partial_off && w_partial_off && tiles && slices && total, "bad arguments");
  BwdLayout L;
  bwd_layout(D, d->feat_cols, rows, groups, &L);
  for (int b = 0; b < D.nl; ++b) {
    ln_partial_off[b] = L.lnp[b];
    w_partial_off[b] = L.wp[b];
  }
  *tiles = L.tiles;
  *slices = L.slices;
  *total = L.total;
  return OCOCC_OK;
}

extern "C" int ococc_sir_layer_fwd_f32(const ococc_sir_layer* d, const float* feats, const float* f_cluster,
                                       const int32_t* inv, int64_t rows, int64_t groups, float* slab, float* y_out,
                                       float* groups_out, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Dims D;
  if (int rc = read_dims(d, &D)) return rc;
  OCOCC_REQUIRE(rows >= 0 && groups >= 0, "negative sizes");
  OCOCC_REQUIRE(feats && inv && slab && y_out && groups_out && (f_cluster || (D.nr == 0 && !d->with_cluster_center)),
                "null pointer");
  FwdLayout L;
  fwd_layout(D, rows, groups, &L);
  const int last = D.nl - 1;
  if (rows > 0 && groups > 0 && sir_fused_enabled()) {   // the whole layer in one launch (csrc/sir_fused.hpp)
    SirFusedArgs A;
    fused_args(d, D, L, feats, f_cluster, inv, rows, groups, slab, y_out, &A);
    A.groups_out = groups_out;
    const int rc = sir_fused_forward(A, stream);
    // (other negative values: not available for this call -- the per-block launches below)
    if (rc >= 0 || rc == OCOCC_ESTRANDED) return rc;
  }
  auto y_of = [&](int b) -> float* { return (b == last && !d->shortcut) ? y_out : slab + L.y[b]; };
  const float* x = f_cluster;
  int ldx = d->cluster_cols;
  for (int j = 0; j < D.nr; ++j) {   // gate = rel_mlp(f_cluster / rel_dist_scaler)
    if (int rc = ococc_point_mlp_fwd_f32(x, D.k[j], ldx, nullptr, 0, j == 0 ? d->rel_colscale : nullptr, nullptr, 0, 0, 1.f,
                                         nullptr, 0, nullptr, rows, d->w_frag[j], D.n[j], d->ln_weight[j], d->ln_bias[j],
                                         d->eps[j], d->act[j], y_of(j), nullptr, 0, stream_))
      return rc;
    x = y_of(j);
    ldx = D.n[j];
  }
  const float* gate = D.nr ? x : d->gate;   // (no rel blocks: the gate may come from outside, [rows, feat_cols])
  for (int i = 0; i < D.nv; ++i) {
    const int q = D.nr + i;
    float* m = slab + L.m[i];
    int rc;
    if (i == 0)
      rc = ococc_point_mlp_fwd_f32(feats, d->feat_cols, d->feat_cols, gate, gate ? d->feat_cols : 0, d->colscale,
                                   d->with_cluster_center ? f_cluster : nullptr, d->with_cluster_center ? d->cluster_cols : 0,
                                   d->with_cluster_center ? d->cluster_cols : 0, d->bscale, nullptr, 0, inv, rows, d->w_frag[q],
                                   D.n[q], d->ln_weight[q], d->ln_bias[q], d->eps[q], d->act[q], y_of(q), m, groups, stream_);
    else
      rc = ococc_point_mlp_fwd_f32(y_of(q - 1), D.n[q - 1], D.n[q - 1], nullptr, 0, nullptr, nullptr, 0, 0, 1.f,
                                   slab + L.m[i - 1], D.n[q - 1], inv, rows, d->w_frag[q], D.n[q], d->ln_weight[q],
                                   d->ln_bias[q], d->eps[q], d->act[q], y_of(q), m, groups, stream_);
    if (rc) return rc;
    // the smallest row attaining each maximum, for the backward pass (either form reads it from the forward slab)
    if (!d->inference && groups > 0)
      if (int rc2 = ococc_point_mlp_segment_argmax(y_of(q), m, inv, rows, D.n[q], groups, (int32_t*)(slab + L.arg[i]), stream_))
        return rc2;
  }
  if (groups > 0) {
    int off = 0;
    for (int i = 0; i < D.nv; ++i) {
      const int n = D.n[D.nr + i];
      hipLaunchKernelGGL(place_cols_kernel, dim3(ococc_grid_1d(groups * n, 256, 2048)), dim3(256), 0, stream,
                         slab + L.m[i], n, groups * n, groups_out, D.sum_n, off);
      off += n;
    }
  }
  if (d->shortcut && rows > 0) {
    const int n = D.n[last];
    hipLaunchKernelGGL(shortcut_kernel, dim3(ococc_grid_1d(rows * n, 256, 4096)), dim3(256), 0, stream, slab + L.y[last],
                       feats, d->feat_cols, n, rows * n, y_out);
  }
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_sir_layer_bwd_f32(const ococc_sir_layer* d, const float* feats, const float* f_cluster,
                                       const int32_t* inv, int64_t rows, int64_t groups, const float* fwd_slab,
                                       const float* y_out, const float* dy, int64_t ld_dy, const float* d_groups,
                                       int64_t ld_dgroups, float* slab, float* dfeat, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Dims D;
  if (int rc = read_dims(d, &D)) return rc;
  OCOCC_REQUIRE(rows >= 0 && groups >= 0, "negative sizes");
  OCOCC_REQUIRE(feats && inv && fwd_slab && slab && (d->shortcut || y_out) && (f_cluster || (D.nr == 0 && !d->with_cluster_center)),
                "null pointer");
  for (int b = 0; b < D.nl; ++b) OCOCC_REQUIRE(d->w_frag[b] && d->wt_frag[b], "weight fragments missing");
  if (rows == 0) return OCOCC_OK;
  FwdLayout F;
  fwd_layout(D, rows, groups, &F);
  BwdLayout L;
  bwd_layout(D, d->feat_cols, rows, groups, &L);
  const int last = D.nl - 1;
  // the gate's gradient: inside the slab when the gate is this layer's rel_mlp, in the caller's buffer when it came from outside
  OCOCC_REQUIRE(D.nr > 0 || !d->gate || d->dgate, "an external gate needs a buffer for its gradient (dgate)");
  float* dgate_buf = (D.nr == 0 && d->gate) ? d->dgate : slab + L.dgate;
  auto y_of = [&](int b) -> const float* { return (b == last && !d->shortcut) ? y_out : fwd_slab + F.y[b]; };
  auto dz_of = [&](int b) -> float* { return slab + L.dz[b]; };
  auto xcat_of = [&](int b) -> float* { return slab + L.xcat[b]; };
  OCOCC_REQUIRE(!d->inference, "the forward pass of this descriptor recorded no arg-max rows (inference = 1)");
  OCOCC_REQUIRE((!dy || ld_dy >= D.n[last]) && (!d_groups || ld_dgroups >= D.sum_n) && ld_dy < (1ll << 31) && ld_dgroups < (1ll << 31),
                "row strides of dy / d_groups shorter than their rows");
  if (groups > 0 && sir_fused_enabled()) {
    SirFusedArgs A;
    fused_args(d, D, F, feats, f_cluster, inv, rows, groups, const_cast<float*>(fwd_slab), const_cast<float*>(y_out), &A);
    for (int b = 0; b < D.nl; ++b) {
      SirBlockArgs& B = A.b[b];
      B.dz = dz_of(b);
      B.xcat = xcat_of(b);
      B.lnp = slab + L.lnp[b];
      B.wp = slab + L.wp[b];
      B.dv = b > D.nr ? slab + L.dv[b - D.nr] : nullptr;
      B.da = (b != 0 && b != D.nr) ? slab + L.dab[b] : nullptr;
    }
    A.dy = dy;
    A.d_groups = d_groups;
    A.ld_dy = (int32_t)ld_dy;
    A.ld_dg = (int32_t)ld_dgroups;
    A.dfeat = dfeat;
    A.dgate = dgate_buf;
    A.slices = L.slices;
    A.rows_per_slice = ococc_align_up(ococc_cdiv(rows, L.slices), 32);
    const int rc = sir_fused_backward(A, stream);
    if (rc > 0 || rc == OCOCC_ESTRANDED) return rc;
    if (rc == 0) {   // dW partials of all blocks: they read every tile's dz and input rows -- a launch of their own
      const float *zs[kMaxBlocks], *xs[kMaxBlocks];
      float* ps[kMaxBlocks];
      for (int b = 0; b < D.nl; ++b) {
        zs[b] = dz_of(b);
        xs[b] = xcat_of(b);
        ps[b] = slab + L.wp[b];
      }
      return ococc_point_mlp_wgrad_multi_f32(D.nl, zs, xs, rows, D.n, D.k, ps, stream_);
    }
  }
  const float* dy_cur = dy;
  if (!dy_cur) {
    OCOCC_HIP(hipMemsetAsync(slab + L.dy0, 0, (size_t)rows * D.n[last] * 4, stream));
    dy_cur = slab + L.dy0;
  } else if (ld_dy != D.n[last]) {   // (the per-block launches read dense rows)
    OCOCC_HIP(hipMemcpy2DAsync(slab + L.dy0, (size_t)D.n[last] * 4, dy, (size_t)ld_dy * 4, (size_t)D.n[last] * 4, (size_t)rows,
                               hipMemcpyDeviceToDevice, stream));
    dy_cur = slab + L.dy0;
  }
  const float* dy_dense = dy ? dy_cur : nullptr;
  const float* carry = nullptr;
  int pp = 0, off_m = D.sum_n;
  for (int i = D.nv - 1; i >= 0; --i) {
    const int q = D.nr + i, n = D.n[q];
    off_m -= n;
    const float* dm = nullptr;
    if (groups > 0 && (d_groups || carry)) {
      if (d_groups) {
        hipLaunchKernelGGL(join_cols_kernel, dim3(ococc_grid_1d(groups * n, 256, 2048)), dim3(256), 0, stream,
                           d_groups + off_m, (int)ld_dgroups, carry, n, groups * n, slab + L.dm);
        dm = slab + L.dm;
      } else {
        dm = carry;
      }
    }
    const int32_t* arg = (const int32_t*)(fwd_slab + F.arg[i]);   // (recorded by the forward pass)
    int rc;
    if (i == 0) {
      const float* gate = D.nr ? y_of(D.nr - 1) : d->gate;
      rc = ococc_point_mlp_bwd_f32(feats, d->feat_cols, d->feat_cols, gate, gate ? d->feat_cols : 0, d->colscale,
                                   d->with_cluster_center ? f_cluster : nullptr, d->with_cluster_center ? d->cluster_cols : 0,
                                   d->with_cluster_center ? d->cluster_cols : 0, d->bscale, nullptr, 0, inv, rows, d->w_frag[q],
                                   d->wt_frag[q], n, d->ln_weight[q], d->ln_bias[q], d->eps[q], d->act[q], dy_cur, dm,
                                   dm ? arg : nullptr, dz_of(q), xcat_of(q), dfeat, gate ? dgate_buf : nullptr, nullptr, nullptr,
                                   slab + L.lnp[q], stream_);
    } else {
      float* dv = slab + L.dv[i];
      OCOCC_HIP(hipMemsetAsync(dv, 0, (size_t)(groups > 0 ? groups : 0) * D.n[q - 1] * 4, stream));
      float* da = slab + L.da[pp];
      rc = ococc_point_mlp_bwd_f32(y_of(q - 1), D.n[q - 1], D.n[q - 1], nullptr, 0, nullptr, nullptr, 0, 0, 1.f,
                                   fwd_slab + F.m[i - 1], D.n[q - 1], inv, rows, d->w_frag[q], d->wt_frag[q], n,
                                   d->ln_weight[q], d->ln_bias[q], d->eps[q], d->act[q], dy_cur, dm, dm ? arg : nullptr,
                                   dz_of(q), xcat_of(q), da, nullptr, nullptr, dv, slab + L.lnp[q], stream_);
      dy_cur = da;
      carry = dv;
      pp ^= 1;
    }
    if (rc) return rc;
  }
  if (d->shortcut && dfeat) {
    const int n = D.n[last];
    const float* dyo = dy_dense ? dy_dense : slab + L.dy0;
    hipLaunchKernelGGL(shortcut_grad_kernel, dim3(ococc_grid_1d(rows * n, 256, 4096)), dim3(256), 0, stream, dfeat,
                       d->feat_cols, dyo, n, rows * n);
  }
  const float* dgate = dgate_buf;
  for (int j = D.nr - 1; j >= 0; --j) {
    const float* x_in = j > 0 ? y_of(j - 1) : f_cluster;
    const int ldx = j > 0 ? D.n[j - 1] : d->cluster_cols;
    float* da = j > 0 ? slab + L.da[pp] : nullptr;
    if (int rc = ococc_point_mlp_bwd_f32(x_in, D.k[j], ldx, nullptr, 0, j == 0 ? d->rel_colscale : nullptr, nullptr, 0, 0,
                                         1.f, nullptr, 0, nullptr, rows, d->w_frag[j], d->wt_frag[j], D.n[j],
                                         d->ln_weight[j], d->ln_bias[j], d->eps[j], d->act[j], dgate, nullptr, nullptr, dz_of(j),
                                         xcat_of(j), da, nullptr, nullptr, nullptr, slab + L.lnp[j], stream_))
      return rc;
    dgate = da;
    pp ^= 1;
  }
  // dW partials of all blocks: one launch
  {
    const float *zs[kMaxBlocks], *xs[kMaxBlocks];
    float* ps[kMaxBlocks];
    for (int b = 0; b < D.nl; ++b) {
      zs[b] = dz_of(b);
      xs[b] = xcat_of(b);
      ps[b] = slab + L.wp[b];
    }
    if (int rc = ococc_point_mlp_wgrad_multi_f32(D.nl, zs, xs, rows, D.n, D.k, ps, stream_)) return rc;
  }
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
