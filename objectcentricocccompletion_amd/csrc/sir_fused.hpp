// A SIRLayer (mmdet3d/models/voxel_encoders/voxel_encoder.py:764-832) as ONE launch per direction: what
// csrc/sir_layer.hip hands to csrc/point_mlp.hip (which holds the kernels: they are the per-block tile bodies of the
// ococc_point_mlp_* launches, run back to back by a persistent grid with a grid-wide barrier wherever the segment maxima
// -- or their gradients -- cross tiles).  Internal to the library: the C ABI stays ococc_sir_layer_{fwd,bwd}_f32.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

constexpr int kSirMaxBlocks = 8;
// per-stream barrier buffer: [0] epoch, [1 + 2 b] count, [2 + 2 b] generation of barrier b; [56..58] the residency census
// (arrived, left, resident together); [62] the epoch target of the last launch in which a wait gave up; [63] which wait (sticky)
constexpr int kSirBarWords = 64, kSirBarError = 63, kSirBarStranded = 62, kSirBarCensus = 56;

struct SirBlockArgs {
  const float* wf;       // weight fragments (ococc_point_mlp_pack_f32)
  const float* wtf;      // ... of the transposed view (backward)
  const float* ln_w;
  const float* ln_b;
  float eps;
  int32_t n, k, act;
  float* y;              // [rows, n] output rows of the block (forward: written; backward: read)
  float* m;              // vfe blocks: segment maxima [groups, n]
  int32_t* arg;          // vfe blocks: smallest row attaining the maximum [groups, n]
  // backward only
  float* dz;             // [rows, n] gradient at the Linear's output
  float* xcat;           // [rows, k] the assembled input
  float* lnp;            // [tiles, 2, n] LayerNorm partial sums
  float* wp;             // [slices, n, k] weight-gradient partial products
  float* dv;             // vfe blocks after the first: gradient of the maxima this block gathered [groups, n of the block before]
  float* da;             // blocks fed by another block: gradient of that block's output rows [rows, its n]
};

struct SirFusedArgs {
  const float* feats;    // [rows, feat_cols]
  const float* fc;       // [rows, cluster_cols]
  const int32_t* inv;    // [rows] non-decreasing
  const float* rel_cs;   // [cluster_cols] or null
  const float* col;      // [feat_cols] or null
  const float* gate;     // nr == 0: an external gate [rows, ld_gate >= feat_cols] multiplied into the features (the rel_mlp
                         // output computed elsewhere: csrc/sir_rel_chains.hip), or null
  int32_t ld_gate;
  int32_t feat_cols, cluster_cols, with_cc, shortcut, nr, nv, sum_n;
  float bscale;
  int64_t rows, groups;
  SirBlockArgs b[kSirMaxBlocks];   // rel blocks first
  float* y_out;          // forward: [rows, n last]
  float* groups_out;     // forward: [groups, sum_n]
  // backward only
  const float* dy;       // [rows, ld_dy >= n last] or null
  const float* d_groups; // [groups, ld_dg >= sum_n] or null
  int32_t ld_dy, ld_dg;  // their row strides in floats
  float* dfeat;          // [rows, feat_cols] or null
  float* dgate;          // [rows, feat_cols]: gradient of the gate (the last rel block's output, or the external gate)
  int64_t rows_per_slice;
  int32_t slices;
  uint32_t* bar;         // grid-barrier words of this stream (sir_fused_barrier_words)
  uint32_t* err_host;    // host-mapped word: which wait gave up (0: none); read by the library before its next launch
  uint64_t bar_ticks;    // bound of a barrier wait in ticks of the 100 MHz clock
  uint32_t* census;      // non-null: the launch only counts how many of its workgroups are resident together
};

// a block signature: which instantiation of the tile bodies every block of the layer runs on (output-channel blocks per
// wave; input-channel blocks per wave in the backward pass).  A layer takes the one-launch form when its blocks are
// exactly a signature's -- the instantiations the per-block launches would pick, so both forms sum in the same order.
struct SirSignature {
  int nr, nv;
  int nbw[kSirMaxBlocks], kbw[kSirMaxBlocks];
};
// The SIRLayers of configs[2] (ococcnet_cfg.py): rel_mlp 3|13 -> 16 -> 32 -> C, vfe C (+3) -> 128, 256 -> 128 with
//   0: C = 131 | 144 (blocks 1..5 of both stacks)     1: C = 15 | 24 (block 0 of either stack)
//   2, 3: the same two without their rel_mlp (the gate comes from outside: all rel_mlps of a stack run as one launch)
constexpr int kSirSignatures = 4;
constexpr SirSignature kSirSignature[kSirSignatures] = {{3, 2, {1, 1, 3, 2, 2}, {1, 1, 1, 3, 4}},
                                                        {3, 2, {1, 1, 1, 2, 2}, {1, 1, 1, 1, 4}},
                                                        {0, 2, {2, 2}, {3, 4}},
                                                        {0, 2, {2, 2}, {1, 4}}};

int point_mlp_tile_rows(int64_t rows);   // csrc/point_mlp.hip
// per tile size (csrc/sir_fused_mb{1,2,4}.hip): set the kernel up for `lds` bytes, size the persistent grid, launch
int sir_fused_launch_mb1(const SirFusedArgs& args, int signature, bool backward, int lds, int64_t tiles, bool one_tile_each,
                         int grid_override, hipStream_t stream);
int sir_fused_launch_mb2(const SirFusedArgs& args, int signature, bool backward, int lds, int64_t tiles, bool one_tile_each,
                         int grid_override, hipStream_t stream);
int sir_fused_launch_mb4(const SirFusedArgs& args, int signature, bool backward, int lds, int64_t tiles, bool one_tile_each,
                         int grid_override, hipStream_t stream);

// 0 = launched; OCOCC_ESTRANDED = a barrier wait of an EARLIER one-launch layer of this process gave up (reported once;
// the per-block launches are pinned from then on); any other negative value = this call cannot take the one-launch form
// (no signature fits the layer; stream under capture with no barrier words or no residency census yet): the caller
// issues the per-block launches instead.
int sir_fused_forward(const SirFusedArgs& args, hipStream_t stream);
int sir_fused_backward(const SirFusedArgs& args, hipStream_t stream);
bool sir_fused_enabled();
