// The rel_mlp chains of SEVERAL SIRLayers in one launch per direction.  A SIRLayer's gate = rel_mlp(f_cluster /
// rel_dist_scaler) (mmdet3d/models/voxel_encoders/voxel_encoder.py:779-790; build_mlp: mmdet3d/ops/sst/sst_ops.py:333-360)
// depends on the cluster offsets only, and the layers of a SIR stack share those (mmdet3d/models/backbones/sir.py:67-88,
// ococc_bbox_head.py:237-316): the chains of all its layers are independent of one another and of the layers' features.
// At the config's own batch (8 k points = 256 row tiles) one chain is three latency-bound bodies on a quarter-filled
// device; six chains side by side are the same three bodies on 1 536 tiles.  Forward: chain c, block j:
// y = act(LN(W x)), x = f_cluster * rel_colscale (j = 0) or the block before; the last y is the gate.  Backward: the same
// tile bodies in reverse from d gate, leaving dz / the input rows / LayerNorm partial rows per block, then the
// weight-gradient products of all blocks in one launch.  The tile bodies are those of csrc/point_mlp_tile.hpp: results
// equal to the bit those of the per-layer launches.
#include "point_mlp_tile.hpp"
#include "sir_fused.hpp"

namespace {

constexpr int kMaxChains = 8, kMaxChainBlocks = 3;

struct RelBlock {
  const float* wf;
  const float* wtf;
  const float* ln_w;
  const float* ln_b;
  float eps;
  int32_t n, k, act;
  float* y;      // [rows, n]
  float* dz;     // backward: [rows, n]
  float* xcat;   // backward: [rows, k]
  float* lnp;    // backward: [tiles, 2, n]
  float* da;     // backward, blocks after the first: [rows, k]
};
struct RelArgs {
  const float* fc;       // [rows, cluster_cols]
  const float* rel_cs[kMaxChains];
  const float* dgate[kMaxChains];   // backward: [rows, n of the chain's last block]
  int32_t cluster_cols, count, n_blocks;
  int32_t wide_last[kMaxChains];    // the last block of the chain has more than 4 output-channel blocks (3 per wave)
  int64_t rows, tiles;
  RelBlock b[kMaxChains][kMaxChainBlocks];
};

using KRel = const __attribute__((address_space(4))) RelArgs;
__device__ __forceinline__ KRel* rel_args() { return (KRel*)__builtin_amdgcn_kernarg_segment_ptr(); }
// (see csrc/sir_fused_impl.hpp: without this the compiler computes every body's addresses in front of the first body)
__device__ __forceinline__ void launder(KRel*& p, Tile& t) {
  asm volatile("" : "+s"(p));
  asm volatile("" : "+v"(t.tid));
}

__device__ __forceinline__ PointMlpIn rel_input(KRel* A, int c, int j) {
  PointMlpIn in{};
  in.rows = A->rows;
  in.bscale = 1.f;
  in.a = j == 0 ? A->fc : A->b[c][j > 0 ? j - 1 : 0].y;
  in.ka = A->b[c][j].k;
  in.lda = j == 0 ? A->cluster_cols : A->b[c][j > 0 ? j - 1 : 0].n;
  in.colscale = j == 0 ? A->rel_cs[c] : nullptr;
  return in;
}

template <int NBW, int MB>
__device__ __forceinline__ void rel_forward_block(KRel* A, int c, int j, Tile t) {
  launder(A, t);
  const PointMlpIn in = rel_input(A, c, j);
  auto& B = A->b[c][j];
  point_mlp_fwd_tile<NBW, MB>(in, B.wf, B.n, B.ln_w, B.ln_b, B.eps, B.act, B.y, nullptr, t);
  __syncthreads();
}

template <int MB>
__global__ void __launch_bounds__(kT, 2) rel_chains_fwd_kernel(RelArgs) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  KRel* A = rel_args();
  const int64_t tiles = A->tiles;
  const int c = (int)((int64_t)blockIdx.x / tiles);
  const int64_t tile = (int64_t)blockIdx.x - c * tiles;
  const Tile t{tile, tile * 16 * MB, smem_f, (int)threadIdx.x};
  const int nb = A->n_blocks;
  for (int j = 0; j + 1 < nb; ++j) rel_forward_block<1, MB>(A, c, j, t);
  if (A->wide_last[c]) rel_forward_block<3, MB>(A, c, nb - 1, t);
  else rel_forward_block<1, MB>(A, c, nb - 1, t);
}

template <int NBW, int MB>
__device__ __forceinline__ void rel_backward_block(KRel* A, int c, int j, const float* dy, Tile t) {
  launder(A, t);
  const PointMlpIn in = rel_input(A, c, j);
  auto& B = A->b[c][j];
  point_mlp_bwd_tile<NBW, 1, MB>(in, B.wf, B.wtf, B.n, B.ln_w, B.ln_b, B.eps, B.act, dy, B.n, nullptr, 0, nullptr, nullptr, B.dz,
                                 B.xcat, j > 0 ? B.da : nullptr, nullptr, nullptr, nullptr, B.lnp, t);
  __syncthreads();
}

template <int MB>
__global__ void __launch_bounds__(kT, 2) rel_chains_bwd_kernel(RelArgs) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  KRel* A = rel_args();
  const int64_t tiles = A->tiles;
  const int c = (int)((int64_t)blockIdx.x / tiles);
  const int64_t tile = (int64_t)blockIdx.x - c * tiles;
  const Tile t{tile, tile * 16 * MB, smem_f, (int)threadIdx.x};
  const int nb = A->n_blocks;
  if (A->wide_last[c]) rel_backward_block<3, MB>(A, c, nb - 1, A->dgate[c], t);
  else rel_backward_block<1, MB>(A, c, nb - 1, A->dgate[c], t);
  for (int j = nb - 2; j >= 0; --j) rel_backward_block<1, MB>(A, c, j, A->b[c][j + 1].da, t);
}

inline int64_t pad64(int64_t v) { return ococc_align_up(v > 0 ? v : 1, 64); }

struct ChainDims {
  int nb, n[kMaxChainBlocks], k[kMaxChainBlocks];
};
inline int read_chain(const ococc_sir_rel_chain* c, ChainDims* D) {
  OCOCC_REQUIRE(c && c->n_blocks >= 1 && c->n_blocks <= kMaxChainBlocks && c->cluster_cols >= 1, "1..3 blocks per chain");
  D->nb = c->n_blocks;
  for (int j = 0; j < D->nb; ++j) {
    OCOCC_REQUIRE(c->n[j] >= 1 && c->n[j] <= 144 && c->w_frag[j] && c->ln_weight[j] && c->ln_bias[j], "bad block");
    D->n[j] = c->n[j];
    D->k[j] = j == 0 ? c->cluster_cols : c->n[j - 1];
    // the instantiations the kernels hold: one output-channel block per wave, three for a wide last block; one
    // input-channel block per wave in the backward pass
    const int nbw = (((D->n[j] + 15) >> 4) + 3) / 4, kbw = (((D->k[j] + 15) >> 4) + 3) / 4;
    if (!(kbw == 1 && (nbw == 1 || (j == D->nb - 1 && nbw == 3)))) return -1;   // (not an error: the caller runs the chain per layer)
  }
  return OCOCC_OK;
}

struct FwdOff {
  int64_t y[kMaxChainBlocks], total;
};
inline void fwd_off(const ChainDims& D, int64_t rows, FwdOff* F) {   // (the last block's y is the gate: the caller's tensor)
  int64_t off = 0;
  for (int j = 0; j + 1 < D.nb; ++j) {
    F->y[j] = off;
    off += pad64(rows * D.n[j]);
  }
  F->total = off > 0 ? off : 64;
}
struct BwdOff {
  int64_t dz[kMaxChainBlocks], xcat[kMaxChainBlocks], da[kMaxChainBlocks], lnp[kMaxChainBlocks], wp[kMaxChainBlocks], total;
  int64_t tiles;
  int slices;
};
inline void bwd_off(const ChainDims& D, int64_t rows, BwdOff* L) {
  L->tiles = ococc_point_mlp_tiles(rows);
  L->slices = ococc_point_mlp_wgrad_slices(rows);
  int64_t off = 0;
  auto take = [&](int64_t n) { int64_t o = off; off += pad64(n); return o; };
  for (int j = 0; j < D.nb; ++j) L->dz[j] = take(rows * D.n[j]);
  for (int j = 0; j < D.nb; ++j) L->xcat[j] = take(rows * D.k[j]);
  for (int j = 0; j < D.nb; ++j) L->da[j] = j > 0 ? take(rows * D.k[j]) : 0;
  for (int j = 0; j < D.nb; ++j) L->lnp[j] = take(L->tiles * 2 * D.n[j]);
  for (int j = 0; j < D.nb; ++j) L->wp[j] = take((int64_t)L->slices * D.n[j] * D.k[j]);
  L->total = off;
}

template <typename K>
int launch_chains(K kernel, const RelArgs& A, int tile_rows, hipStream_t stream) {
  int lds_floats_max = 0;
  for (int c = 0; c < A.count; ++c)
    for (int j = 0; j < A.n_blocks; ++j) {
      const int f = lds_floats(pad_k(A.b[c][j].k), pad_k(A.b[c][j].n), tile_rows);
      lds_floats_max = f > lds_floats_max ? f : lds_floats_max;
    }
  const int lds = lds_floats_max * 4;
  OCOCC_REQUIRE(lds <= 64 * 1024, "tile too large");   // (rel blocks: <= 144 x 32 -- a quarter of that)
  hipLaunchKernelGGL(kernel, dim3((unsigned)(A.count * A.tiles)), dim3(kT), lds, stream, A);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

}  // namespace

extern "C" int64_t ococc_sir_rel_chain_fwd_floats(const ococc_sir_rel_chain* chain, int64_t rows) {
  ChainDims D;
  if (rows < 0 || read_chain(chain, &D) != OCOCC_OK) return -1;
  FwdOff F;
  fwd_off(D, rows, &F);
  return F.total;
}

extern "C" int ococc_sir_rel_chain_bwd_layout(const ococc_sir_rel_chain* chain, int64_t rows, int64_t* ln_partial_off,
                                              int64_t* w_partial_off, int64_t* tiles, int32_t* slices, int64_t* total_floats) {
  ChainDims D;
  const int rc = read_chain(chain, &D);
  OCOCC_REQUIRE(rc == OCOCC_OK, "this chain's block widths are outside the batched kernels (ococc_sir_rel_chain_fwd_floats < 0)");
  OCOCC_REQUIRE(rows >= 0 && ln_partial_off && w_partial_off && tiles && slices && total_floats, "bad arguments");
  BwdOff L;
  bwd_off(D, rows, &L);
  for (int j = 0; j < D.nb; ++j) {
    ln_partial_off[j] = L.lnp[j];
    w_partial_off[j] = L.wp[j];
  }
  *tiles = L.tiles;
  *slices = L.slices;
  *total_floats = L.total;
  return OCOCC_OK;
}

static int fill_args(int32_t count, const ococc_sir_rel_chain* chains, const float* f_cluster, int64_t rows, RelArgs* A,
                     ChainDims* D) {
  OCOCC_REQUIRE(count >= 1 && count <= kMaxChains && chains && f_cluster && rows >= 0, "1..8 chains per call");
  *A = RelArgs{};
  A->fc = f_cluster;
  A->cluster_cols = chains[0].cluster_cols;
  A->count = count;
  A->n_blocks = chains[0].n_blocks;
  A->rows = rows;
  for (int c = 0; c < count; ++c) {
    const int rc = read_chain(&chains[c], &D[c]);
    OCOCC_REQUIRE(rc == OCOCC_OK, "a chain's block widths are outside the batched kernels");
    OCOCC_REQUIRE(chains[c].n_blocks == A->n_blocks && chains[c].cluster_cols == A->cluster_cols,
                  "the chains of one call have the same depth and input");
    A->rel_cs[c] = chains[c].rel_colscale;
    A->wide_last[c] = (((D[c].n[D[c].nb - 1] + 15) >> 4) + 3) / 4 == 3;
    for (int j = 0; j < D[c].nb; ++j) {
      RelBlock& B = A->b[c][j];
      B.wf = chains[c].w_frag[j];
      B.wtf = chains[c].wt_frag[j];
      B.ln_w = chains[c].ln_weight[j];
      B.ln_b = chains[c].ln_bias[j];
      B.eps = chains[c].eps[j];
      B.n = D[c].n[j];
      B.k = D[c].k[j];
      B.act = chains[c].act[j];
    }
  }
  return OCOCC_OK;
}

extern "C" int ococc_sir_rel_chains_fwd_f32(int32_t count, const ococc_sir_rel_chain* chains, const float* f_cluster, int64_t rows,
                                            float* const* slabs, float* const* gates, ococc_stream_t stream_) {
  RelArgs A;
  ChainDims D[kMaxChains];
  if (int rc = fill_args(count, chains, f_cluster, rows, &A, D)) return rc;
  OCOCC_REQUIRE(slabs && gates, "null pointer table");
  if (rows == 0) return OCOCC_OK;
  for (int c = 0; c < count; ++c) {
    OCOCC_REQUIRE(slabs[c] && gates[c], "null pointer");
    FwdOff F;
    fwd_off(D[c], rows, &F);
    for (int j = 0; j < D[c].nb; ++j) A.b[c][j].y = j + 1 < D[c].nb ? slabs[c] + F.y[j] : gates[c];
  }
  const int tile_rows = point_mlp_tile_rows(rows);
  A.tiles = ococc_cdiv(rows, tile_rows);
  if (tile_rows == 16) return launch_chains(rel_chains_fwd_kernel<1>, A, tile_rows, (hipStream_t)stream_);
  if (tile_rows == 32) return launch_chains(rel_chains_fwd_kernel<2>, A, tile_rows, (hipStream_t)stream_);
  return launch_chains(rel_chains_fwd_kernel<4>, A, tile_rows, (hipStream_t)stream_);
}

extern "C" int ococc_sir_rel_chains_bwd_f32(int32_t count, const ococc_sir_rel_chain* chains, const float* f_cluster, int64_t rows,
                                            const float* const* fwd_slabs, const float* const* gates,
                                            const float* const* dgates, float* const* slabs, ococc_stream_t stream_) {
  RelArgs A;
  ChainDims D[kMaxChains];
  if (int rc = fill_args(count, chains, f_cluster, rows, &A, D)) return rc;
  OCOCC_REQUIRE(fwd_slabs && gates && dgates && slabs, "null pointer table");
  if (rows == 0) return OCOCC_OK;
  const float *zs[kMaxChains * kMaxChainBlocks], *xs[kMaxChains * kMaxChainBlocks];
  float* ps[kMaxChains * kMaxChainBlocks];
  int32_t ns[kMaxChains * kMaxChainBlocks], ks[kMaxChains * kMaxChainBlocks];
  int nw = 0;
  for (int c = 0; c < count; ++c) {
    OCOCC_REQUIRE(fwd_slabs[c] && gates[c] && dgates[c] && slabs[c], "null pointer");
    for (int j = 0; j < D[c].nb; ++j) OCOCC_REQUIRE(chains[c].wt_frag[j], "transposed weight fragments missing");
    FwdOff F;
    fwd_off(D[c], rows, &F);
    BwdOff L;
    bwd_off(D[c], rows, &L);
    A.dgate[c] = dgates[c];
    for (int j = 0; j < D[c].nb; ++j) {
      RelBlock& B = A.b[c][j];
      B.y = j + 1 < D[c].nb ? const_cast<float*>(fwd_slabs[c]) + F.y[j] : const_cast<float*>(gates[c]);
      B.dz = slabs[c] + L.dz[j];
      B.xcat = slabs[c] + L.xcat[j];
      B.lnp = slabs[c] + L.lnp[j];
      B.da = j > 0 ? slabs[c] + L.da[j] : nullptr;
      zs[nw] = B.dz;
      xs[nw] = B.xcat;
      ps[nw] = slabs[c] + L.wp[j];
      ns[nw] = B.n;
      ks[nw] = B.k;
      ++nw;
    }
  }
  const int tile_rows = point_mlp_tile_rows(rows);
  A.tiles = ococc_cdiv(rows, tile_rows);
  int rc;
  if (tile_rows == 16) rc = launch_chains(rel_chains_bwd_kernel<1>, A, tile_rows, (hipStream_t)stream_);
  else if (tile_rows == 32) rc = launch_chains(rel_chains_bwd_kernel<2>, A, tile_rows, (hipStream_t)stream_);
  else rc = launch_chains(rel_chains_bwd_kernel<4>, A, tile_rows, (hipStream_t)stream_);
  if (rc) return rc;
  return ococc_point_mlp_wgrad_multi_f32(nw, zs, xs, rows, ns, ks, ps, stream_);   // dW partials of every block: one launch
}
