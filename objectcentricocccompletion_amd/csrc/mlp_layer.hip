// A11 / A10, fused: one layer of the occupancy decoder's per-query MLP,  y = dropout(GELU(LayerNorm(x W^T + g[idx]))),
// over ~1e6 query rows (OccDecoder.forward, mmdet3d/models/occ/occ_base.py:99-153: build_mlp's Sequential(Linear(bias=
// False), LN(eps 1e-3), GELU, Dropout(0.1)) blocks 1596 -> 512 -> 1024 -> 1024 and the Linear(1024 -> 1) head,
// mmdet3d/ops/sst/sst_ops.py:333-360).  bf16 operands, f32 accumulation / statistics.
//
// The library GEMMs reach 21-25 % of the bf16 peak on these shapes and every layer is followed by a LayerNorm kernel that
// reads and writes the [1 M, 1024] activation again (6 ms of a 79 ms step).  Here a workgroup of 8 waves owns 64 query
// rows and ALL N <= 1024 output channels, so that the LayerNorm over a row is an exchange inside the workgroup:
//   * x tile [64][K] bf16 in LDS (row stride K + 16 elements: the 32 rows of an MFMA operand read hit different banks);
//     the NEXT tile's rows are copied global -> LDS by DMA (global_load_lds_dwordx4, no registers) while this tile's
//     epilogue runs;
//   * out^T = W x^T on v_mfma_f32_32x32x16_bf16 (32 cycles per instruction and SIMD; the same FLOP per clock as the
//     16x16x32 shape at 16 -- tools/probe/mfma_rate.hip, corrected in round 4 -- with half the A-operand fragments per
//     FLOP, which is what this L2-bound kernel is short of) -- weights as the A operand in fragment order (one wave load = 1 KB
//     of consecutive bytes, ococc_linear_fragments32_bf16), streamed from L2 through a 4-deep register ring by inline
//     asm loads with hand-counted waits (the compiler sinks plain loads to their uses, csrc/point_mlp.hip);
//   * a wave owns N / 8 channels of all 64 rows: 4 (or 2) x 2 accumulator tiles of 32 x 32;
//   * epilogue in registers: + g[idx[row]] (the per-RoI half of the factorised first layer), LayerNorm statistics
//     across the 8 waves through LDS, GELU, the counter-based dropout mask of csrc/ln_math.hpp, bf16 stores of 4
//     consecutive channels; optionally the 1024 -> 1 head as a dot product over the activated row.
// Weight traffic: N K 2 bytes per 64 rows from L2 (3 MB for the two wide layers = 50 GB per 1 M queries): at the
// matrix pipe's rate the L2 has to deliver ~4.8 TB/s per XCD, which is its limit -- the kernel is L2-bound by design.
#include "common.hpp"
#include "ln_math.hpp"

namespace {

constexpr int TM = 64;
constexpr int kThreads = 512;
constexpr int kWaves = 8;
constexpr int kRing = 4;      // k-steps of weight fragments in flight
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct MlpLayerArgs {
  const uint16_t* x;        // [rows, K] bf16
  const float* add;         // [*, N] f32 or null
  const int32_t* add_idx;   // [rows] or null
  const uint16_t* wfrag;    // [N/32][K/16][64][8] bf16
  const float* ln_w;        // [N] (null: no LayerNorm)
  const float* ln_b;
  const float* bias;        // [N] or null (added before the LayerNorm)
  uint16_t* y;              // [rows, N] bf16 or null
  const float* head_w;      // [N] or null
  float* head_out;          // [rows]
  const float* head_b;      // one float or null
  float eps;
  int32_t act;              // 0 none, 1 gelu
  int32_t K, N;
  int64_t rows;
  LnDropout drop;
};

template <int OFF>   // (the k-step inside a ring round is an immediate offset: one 64-bit base per channel block)
__device__ __forceinline__ void frag_load16(bf16x8& dst, uint32_t lane_off, const void* base) {
  // wave-uniform base in scalar registers + one per-lane offset register: no 64-bit address registers per block
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_off), "s"(base), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
}
__device__ __forceinline__ void frag_tie(bf16x8& v) { asm volatile("" : "+v"(v)); }

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// 64 lanes x 16 bytes, global -> LDS at lds_addr + 16 * lane (LDS base in M0), no register hop
__device__ __forceinline__ void dma_b128(const void* src, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop

// rows [row0, row0 + 64) of x -> the LDS tile (row stride ld elements).  K * 2 bytes a multiple of 1 KB: DMA, a wave per 8
// rows; otherwise (the 64-wide positional-encoding input) through registers.  Rows past the end re-read the last row.
__device__ __forceinline__ void load_x_tile(const MlpLayerArgs& a, int64_t row0, uint16_t* xs, int ld) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row_bytes = a.K * 2;
  if ((row_bytes & 1023) == 0) {
    const int chunks = row_bytes >> 10;
    for (int rr = 0; rr < 8; ++rr) {
      const int r = 8 * wave + rr;
      const int64_t row = row0 + r < a.rows ? row0 + r : a.rows - 1;
      const char* src = (const char*)a.x + row * row_bytes + lane * 16;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(   // LDS byte address of the row
          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(xs + r * ld));
      for (int ch = 0; ch < chunks; ++ch) dma_b128(src + ch * 1024, dst + ch * 1024);
    }
  } else {
    const int pieces = row_bytes >> 4;   // 16-byte pieces per row
    for (int i = threadIdx.x; i < TM * pieces; i += kThreads) {
      const int r = i / pieces, p = i - r * pieces;
      const int64_t row = row0 + r < a.rows ? row0 + r : a.rows - 1;
      *(u32x4*)(xs + r * ld + p * 8) = *(const u32x4*)((const char*)a.x + row * row_bytes + p * 16);
    }
  }
}

// ---- the GEMM phase shared by the kernels below: out^T [32 NPW channels of the wave][64 rows] += W x^T over KS k-steps.
// Slot S of the ring holds the fragments of k-step kRing * round + S; the k-step inside a round is an immediate offset.
template <int S, int NPW>
__device__ __forceinline__ void ring_issue(bf16x8 (&ring)[kRing][NPW], uint32_t lane_off, const char* const (&wp)[NPW],
                                           int round) {
#pragma unroll
  for (int nb = 0; nb < NPW; ++nb) frag_load16<S * 1024>(ring[S][nb], lane_off, wp[nb] + (size_t)round * (kRing * 1024));
}
template <int S, int NPW>
__device__ __forceinline__ void ring_step(f32x16 (&acc)[NPW][2], bf16x8 (&ring)[kRing][NPW], bf16x8 (&b)[2], const uint16_t* xb,
                                          int ld, int next_step, int next_round, uint32_t lane_off,
                                          const char* const (&wp)[NPW]) {
  vm_wait<(kRing - 1) * NPW>();   // the three younger slots may still be out, this slot's loads have landed
#pragma unroll
  for (int nb = 0; nb < NPW; ++nb) frag_tie(ring[S][nb]);
  const bf16x8 b0 = b[0], b1 = b[1];
  // the NEXT step's token operands leave LDS while this step's products run (past the end: the last step's again)
  b[0] = *(const bf16x8*)(xb + next_step * 16);
  b[1] = *(const bf16x8*)(xb + 32 * ld + next_step * 16);
#pragma unroll
  for (int nb = 0; nb < NPW; ++nb) {
    acc[nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[S][nb], b0, acc[nb][0], 0, 0, 0);
    acc[nb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[S][nb], b1, acc[nb][1], 0, 0, 0);
  }
  ring_issue<S>(ring, lane_off, wp, next_round);   // (always issued -- past the end the last round again -- so that the
}                                                  //  wait counts stay uniform)
// No fragment load is in flight on entry or on return: the ring's registers hold nothing the compiler does not know
// about while other code (the epilogue, with its spills) runs.  The price is one exposed L2 round trip per phase.
template <int NPW>
__device__ __forceinline__ void gemm_rows64(f32x16 (&acc)[NPW][2], const uint16_t* xs, int ld, int ks, const uint16_t* wfrag,
                                            int tid_) {
  const int lane = tid_ & 63, m = lane & 31, h = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const char* wp[NPW];   // wave-uniform: scalar registers
#pragma unroll
  for (int nb = 0; nb < NPW; ++nb) wp[nb] = (const char*)wfrag + (size_t)(wave_u * NPW + nb) * ks * 1024;
  const uint32_t lane_off = lane * 16;
  const int last_round = ks / kRing - 1;
  bf16x8 ring[kRing][NPW];
  ring_issue<0>(ring, lane_off, wp, 0);
  ring_issue<1>(ring, lane_off, wp, 0);
  ring_issue<2>(ring, lane_off, wp, 0);
  ring_issue<3>(ring, lane_off, wp, 0);
#pragma unroll
  for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][mb][i] = 0.f;
  const uint16_t* xb = xs + m * ld + 8 * h;
  bf16x8 b[2] = {*(const bf16x8*)xb, *(const bf16x8*)(xb + 32 * ld)};
#pragma unroll 1
  for (int round = 0; round <= last_round; ++round) {
    const int nxt = round < last_round ? round + 1 : last_round;
    const int k0 = kRing * round;
    ring_step<0>(acc, ring, b, xb, ld, k0 + 1, nxt, lane_off, wp);
    ring_step<1>(acc, ring, b, xb, ld, k0 + 2, nxt, lane_off, wp);
    ring_step<2>(acc, ring, b, xb, ld, k0 + 3, nxt, lane_off, wp);
    ring_step<3>(acc, ring, b, xb, ld, round < last_round ? k0 + 4 : k0 + 3, nxt, lane_off, wp);
  }
  vm_wait<0>();   // the clamped requests past the end: landed, discarded
#pragma unroll
  for (int s = 0; s < kRing; ++s)
#pragma unroll
    for (int nb = 0; nb < NPW; ++nb) frag_tie(ring[s][nb]);
}

// ---- the epilogue shared by the kernels below.  Register i of block nb, row block mb holds
//   channel 32 (NPW wave + nb) + (i & 3) + 8 (i >> 2) + 4 h  of row 32 mb + m        (m = lane & 31, h = lane >> 5)
// rows of an LDS tile -> global, full rows per wave instruction group (defined with the whole-MLP kernel below)
template <int N>
__device__ __forceinline__ void copy_tile_out(const uint16_t* ys, int ld, uint16_t* dst, int64_t row0, int64_t rows);

struct Epilogue {
  const float* gam_s;       // LDS, [N] each; bias_s / head_s null: absent
  const float* bet_s;
  const float* bias_s;
  const float* head_s;
  const float* add;         // global f32 [*, N] rows gathered by add_idx (template flag ADD)
  const int32_t* add_idx;
  float* red0;              // LDS [8 waves][64 rows] each
  float* red1;
  uint16_t* y_global;       // bf16 [rows, N] or null
  uint16_t* z_global;       // TRAIN: bf16 [rows, N], the LayerNorm input as the backward pass reads it
  float* stats_global;      // TRAIN: f32 [rows, 2] (mean, rstd)
  uint16_t* stage;          // TRAIN: LDS tile [64][stage_ld] z is staged through on its way out (the tile the GEMM phase
  int32_t stage_ld;         //        read: free behind the first LayerNorm barrier)
  uint16_t* y_lds;          // bf16 tile [64][ld_lds] or null
  float* head_out;          // f32 [rows] (with head_s)
  const float* head_b;
  int64_t row0, rows;
  int32_t ld_lds;
  int32_t has_ln, act;
  float eps;
  LnDropout drop;
};
// Contains workgroup barriers when has_ln or head_s (kernel-uniform).  Nothing is written to y_lds before the first of
// them, so with a LayerNorm the tile the GEMM phase read may be the one y_lds overwrites.
// TRAIN: what a backward pass needs is written out as well -- z rounded to bf16 (and the LayerNorm then works on the
// ROUNDED values, as it does when a library GEMM hands it a bf16 tensor: forward and backward see the same numbers)
// and the row statistics.
// STORE (default: with TRAIN): z / statistics leave for a backward pass that reads them; TRAIN without STORE is the forward
// of the recomputing backward (occ_mlp_bwd_kernel below): the same numbers, nothing kept.
// STORE 2: for the one-launch backward WITHOUT recompute (occ_mlp_bwd_kernel<.., false>): z is parked in the lane layout
// that kernel reads (z_global: this tile's strip, [group][thread] pairs of bf16 pairs -- straight from the registers, no
// staging through the tile) beside the statistics; y2 is not kept.
template <int NPW, bool ADD, bool MAY_DROP = true, bool TRAIN = false, int STORE = TRAIN ? 1 : 0>
__device__ __forceinline__ void layer_epilogue(f32x16 (&acc)[NPW][2], const Epilogue& e) {
  constexpr int N = NPW * 32 * kWaves;
  // The per-lane indices are loop invariant in the persistent kernels; left alone the compiler computes every one of them
  // ahead of the tile loop and then spills them around the accumulators.  Re-derive them from an opaque thread index.
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int m = tid_ & 31, h = (tid_ >> 5) & 1, wave = tid_ >> 6, nb0 = wave * NPW;
  int64_t rows_of[2];
  int32_t gidx[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    rows_of[mb] = e.row0 + 32 * mb + m;
    gidx[mb] = (ADD && rows_of[mb] < e.rows) ? e.add_idx[rows_of[mb]] : 0;
  }
  if (ADD || e.bias_s) {
#pragma unroll
    for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = 32 * (nb0 + nb) + 8 * q + 4 * h;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (e.bias_s) bias = *(const f32x4*)(e.bias_s + n);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          f32x4 g = bias;
          if (ADD) g = g + *(const f32x4*)(e.add + (int64_t)gidx[mb] * N + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[nb][mb][4 * q + r] += g[r];
        }
        // keep the scheduler from hoisting every block's gathers to the top: with the accumulators live there is room
        // for one block's worth
        if (q == 3) __builtin_amdgcn_sched_barrier(0);
      }
  }
  if (TRAIN) {
#pragma unroll
    for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          u32x2 o;
          o.x = ln_pack2(ln_f32x2{acc[nb][mb][4 * q], acc[nb][mb][4 * q + 1]});
          o.y = ln_pack2(ln_f32x2{acc[nb][mb][4 * q + 2], acc[nb][mb][4 * q + 3]});
          acc[nb][mb][4 * q] = __uint_as_float(o.x << 16);
          acc[nb][mb][4 * q + 1] = __uint_as_float(o.x & 0xffff0000u);
          acc[nb][mb][4 * q + 2] = __uint_as_float(o.y << 16);
          acc[nb][mb][4 * q + 3] = __uint_as_float(o.y & 0xffff0000u);
          if (STORE == 2) *(u32x2*)((uint32_t*)e.z_global + ((((nb * 4 + q) * 2 + mb) * kThreads + tid_) << 1)) = o;
        }
      }
  }
  float rstd[2] = {1.f, 1.f};
  if (e.has_ln) {
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[mb] += acc[nb][mb][i];
      s[mb] += __shfl_xor(s[mb], 32, 64);
      if (h == 0) e.red0[wave * TM + 32 * mb + m] = s[mb];
    }
    __syncthreads();
    if (STORE == 1) {
      // every wave is past its GEMM reads of the tile: z (the accumulators hold bf16 values) goes into it, and leaves
      // behind the second barrier as full rows -- written from here, 8 bytes per lane and 32 rows per instruction, the
      // three z tensors and y2 cost more than the whole arithmetic
#pragma unroll
      for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = 32 * (nb0 + nb) + 8 * q + 4 * h;
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            u32x2 o;
            o.x = (__float_as_uint(acc[nb][mb][4 * q]) >> 16) | (__float_as_uint(acc[nb][mb][4 * q + 1]) & 0xffff0000u);
            o.y = (__float_as_uint(acc[nb][mb][4 * q + 2]) >> 16) | (__float_as_uint(acc[nb][mb][4 * q + 3]) & 0xffff0000u);
            *(u32x2*)(e.stage + (32 * mb + m) * e.stage_ld + n) = o;
          }
        }
    }
    float q2[2], mean_[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) t += e.red0[w * TM + 32 * mb + m];
      const float mean = t / (float)N;
      mean_[mb] = mean;
      q2[mb] = 0.f;
#pragma unroll
      for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[nb][mb][i] -= mean;
          q2[mb] += acc[nb][mb][i] * acc[nb][mb][i];
        }
      q2[mb] += __shfl_xor(q2[mb], 32, 64);
      if (h == 0) e.red1[wave * TM + 32 * mb + m] = q2[mb];
    }
    __syncthreads();
    if (STORE == 1) {
      copy_tile_out<N>(e.stage, e.stage_ld, e.z_global, e.row0, e.rows);
      __syncthreads();   // (the tile receives y next)
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) t += e.red1[w * TM + 32 * mb + m];
      rstd[mb] = rsqrtf(t / (float)N + e.eps);
      if (STORE && wave == 0 && h == 0 && rows_of[mb] < e.rows) {
        e.stats_global[rows_of[mb] * 2] = mean_[mb];
        e.stats_global[rows_of[mb] * 2 + 1] = rstd[mb];
      }
    }
  }
  float dot[2] = {0.f, 0.f};
#pragma unroll
  for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = 32 * (nb0 + nb) + 8 * q + 4 * h;
      f32x4 gm = {1.f, 1.f, 1.f, 1.f}, bt = {0.f, 0.f, 0.f, 0.f}, hw = {0.f, 0.f, 0.f, 0.f};
      if (e.has_ln) {
        gm = *(const f32x4*)(e.gam_s + n);
        bt = *(const f32x4*)(e.bet_s + n);
      }
      if (e.head_s) hw = *(const f32x4*)(e.head_s + n);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        ln_f32x2 v0 = {acc[nb][mb][4 * q] * rstd[mb] * gm[0] + bt[0], acc[nb][mb][4 * q + 1] * rstd[mb] * gm[1] + bt[1]};
        ln_f32x2 v1 = {acc[nb][mb][4 * q + 2] * rstd[mb] * gm[2] + bt[2], acc[nb][mb][4 * q + 3] * rstd[mb] * gm[3] + bt[3]};
        if (e.act == 1) {
          v0 = ln_gelu2(v0);
          v1 = ln_gelu2(v1);
        }
        if (MAY_DROP && e.drop.thr) {
          v0 = v0 * ln_dropout_mask2(e.drop, rows_of[mb], n >> 1, N >> 1);
          v1 = v1 * ln_dropout_mask2(e.drop, rows_of[mb], (n >> 1) + 1, N >> 1);
        }
        u32x2 o;
        o.x = ln_pack2(v0);
        o.y = ln_pack2(v1);
        if (e.head_s) {   // the head reads the activation as the next Linear would: bf16
          dot[mb] += __uint_as_float(o.x << 16) * hw[0] + __uint_as_float(o.x & 0xffff0000u) * hw[1] +
                     __uint_as_float(o.y << 16) * hw[2] + __uint_as_float(o.y & 0xffff0000u) * hw[3];
        }
        if (e.y_lds) *(u32x2*)(e.y_lds + (32 * mb + m) * e.ld_lds + n) = o;
        if (e.y_global && rows_of[mb] < e.rows) *(u32x2*)(e.y_global + rows_of[mb] * N + n) = o;
      }
      if (q & 1) __builtin_amdgcn_sched_barrier(0);   // (two 8-channel pieces per scheduling region: their chains interleave)
    }
  if (e.head_s) {
    __syncthreads();   // (red0 was last read behind the second LayerNorm barrier)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      float d = dot[mb] + __shfl_xor(dot[mb], 32, 64);
      if (h == 0) e.red0[wave * TM + 32 * mb + m] = d;
    }
    __syncthreads();
    if (tid_ < TM && e.row0 + tid_ < e.rows) {
      float t = e.head_b ? e.head_b[0] : 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) t += e.red0[w * TM + tid_];
      e.head_out[e.row0 + tid_] = t;
    }
  }
}

// ---- one layer per launch
template <int NPW, bool ADD>
__global__ void __launch_bounds__(kThreads, 2)
mlp_layer_fwd_kernel(MlpLayerArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int K = a.K, N = a.N, ld = K + 16, KS = K >> 4;
  uint16_t* xs = (uint16_t*)smem;
  float* red0 = (float*)(xs + TM * ld);          // [8 waves][64 rows]
  float* red1 = red0 + kWaves * TM;
  // per-channel parameters on chip: the epilogue then issues no global load that would queue up behind (and so wait
  // for) the DMA of the next tile's rows
  float* gam_s = red1 + kWaves * TM;             // [N] each: LayerNorm weight, bias, Linear bias, head weight
  float* bet_s = gam_s + N;
  float* bias_s = bet_s + N;
  float* head_s = bias_s + N;
  for (int i = threadIdx.x; i < N; i += kThreads) {
    gam_s[i] = a.ln_w ? a.ln_w[i] : 1.f;
    bet_s[i] = a.ln_b ? a.ln_b[i] : 0.f;
    bias_s[i] = a.bias ? a.bias[i] : 0.f;
    head_s[i] = a.head_w ? a.head_w[i] : 0.f;
  }
  const int64_t tiles = (a.rows + TM - 1) / TM;
  Epilogue e;
  e.z_global = nullptr;
  e.stats_global = nullptr;
  e.stage = nullptr;
  e.stage_ld = 0;
  e.gam_s = gam_s;
  e.bet_s = bet_s;
  e.bias_s = a.bias ? bias_s : nullptr;
  e.head_s = a.head_w ? head_s : nullptr;
  e.add = a.add;
  e.add_idx = a.add_idx;
  e.red0 = red0;
  e.red1 = red1;
  e.y_global = a.y;
  e.y_lds = nullptr;
  e.ld_lds = 0;
  e.head_out = a.head_out;
  e.head_b = a.head_b;
  e.rows = a.rows;
  e.has_ln = a.ln_w != nullptr;
  e.act = a.act;
  e.eps = a.eps;
  e.drop = a.drop;
  int64_t tile = blockIdx.x;
  if (tile < tiles) load_x_tile(a, tile * TM, xs, ld);
#pragma unroll 1
  for (; tile < tiles; tile += gridDim.x) {
    vm_wait<0>();      // this tile's rows (DMA, requested before the previous tile's epilogue) have landed
    __syncthreads();
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    f32x16 acc[NPW][2];
    gemm_rows64<NPW>(acc, xs, ld, KS, a.wfrag, tid_);
    __syncthreads();   // every wave has read its last B operands: xs is free for the next tile's rows
    if (tile + gridDim.x < tiles) load_x_tile(a, (tile + gridDim.x) * TM, xs, ld);
    e.row0 = tile * TM;
    layer_epilogue<NPW, ADD>(acc, e);
  }
  vm_wait<0>();   // nothing stays in flight into the LDS of a finished workgroup
}

// ---- the whole decoder MLP in one launch: 64 -> 512 -> 1024 -> 1024 -> 1.  A 64-row tile's activations stay in LDS
// between the layers: y1 [64][1040] bf16 takes 130 KB, y0 [64][528] sits in its first half (dead once y1 is written, which
// happens behind the LayerNorm barriers of layer 1's epilogue, i.e. after every wave's last read of y0) and the
// positional-encoding tile [64][80] in its second half.  Per 64 query rows the kernel reads 128 B of positional encoding
// per row and 3.06 MB of weights from L2, and writes 4 B per row.
constexpr int kN0 = 512, kN1 = 1024, kN2 = 1024, kK0 = 64;
constexpr int kLd0 = kK0 + 16, kLd1 = kN0 + 16, kLd2 = kN1 + 16;
constexpr int kPeOffset = TM * kLd1;                      // (elements) behind the y0 tile
static_assert(kPeOffset + TM * kLd0 <= TM * kLd2, "the positional-encoding tile must fit behind y0 inside y1");
constexpr int kOccMlpLds = TM * kLd2 * 2 + 2 * kWaves * TM * 4 + (2 * (kN0 + kN1 + kN2) + kN2) * 4;
static_assert(kOccMlpLds <= 160 * 1024, "LDS budget");
struct OccMlpArgs {
  const uint16_t* pe;       // [rows, 64] bf16 (ococc_pos_encode_bf16)
  const float* add;         // [*, 512] f32: the per-RoI half of the first layer
  const int32_t* add_idx;   // [rows]
  const uint16_t* w[3];     // fragments of [512, 64], [1024, 512], [1024, 1024]
  const float* ln_w[3];
  const float* ln_b[3];
  const float* head_w;      // [1024]
  const float* head_b;      // [1] or null
  float* out;               // [rows]
  uint16_t* y_out[2];       // optional copies of y0 [rows, 512], y1 [rows, 1024] (what a backward pass starts from)
  uint16_t* y2_out;         // TRAIN: y2 [rows, 1024]
  uint16_t* z_out[3];       // TRAIN: LayerNorm inputs (bf16) ...
  float* stats_out[3];      // ... and row statistics [rows, 2] of the three layers
  float eps;
  int64_t rows;
  LnDropout drop[3];
};
// rows of an LDS tile -> global, full 1 KB (or 2 KB) rows per wave instruction group
template <int N>
__device__ __forceinline__ void copy_tile_out(const uint16_t* ys, int ld, uint16_t* dst, int64_t row0, int64_t rows) {
  constexpr int pieces = N / 8;   // 16-byte pieces per row
  // (an opaque thread index: the piece coordinates are invariant across the persistent kernels' tile loops, and left alone
  // the compiler computes every call's set ahead of the loop and keeps them in registers through all of its phases)
  int t0 = threadIdx.x;
  asm volatile("" : "+v"(t0));
  for (int i = t0; i < TM * pieces; i += kThreads) {
    const int r = i / pieces, p = i - r * pieces;
    if (row0 + r < rows) *(u32x4*)(dst + (row0 + r) * N + p * 8) = *(const u32x4*)(ys + r * ld + p * 8);
  }
}
template <bool DROP, bool TRAIN = false, int STORE = TRAIN ? 1 : 0>   // (inference instantiation: no dropout test per channel pair in the epilogues)
__global__ void __launch_bounds__(kThreads, 2)
occ_mlp_fwd_kernel(OccMlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* ys = (uint16_t*)smem;                 // y1 [64][kLd2]; y0 [64][kLd1] at its start
  uint16_t* ps = ys + kPeOffset;                  // positional-encoding tile [64][kLd0]
  float* red0 = (float*)(ys + TM * kLd2);
  float* red1 = red0 + kWaves * TM;
  float* gam0 = red1 + kWaves * TM;
  float* bet0 = gam0 + kN0;
  float* gam1 = bet0 + kN0;
  float* bet1 = gam1 + kN1;
  float* gam2 = bet1 + kN1;
  float* bet2 = gam2 + kN2;
  float* head_s = bet2 + kN2;
  for (int i = threadIdx.x; i < kN1; i += kThreads) {
    if (i < kN0) {
      gam0[i] = a.ln_w[0][i];
      bet0[i] = a.ln_b[0][i];
    }
    gam1[i] = a.ln_w[1][i];
    bet1[i] = a.ln_b[1][i];
    gam2[i] = a.ln_w[2][i];
    bet2[i] = a.ln_b[2][i];
    head_s[i] = a.head_w[i];
  }
  const int64_t tiles = (a.rows + TM - 1) / TM;
  Epilogue e;
  e.bias_s = nullptr;
  e.add = a.add;
  e.add_idx = a.add_idx;
  e.red0 = red0;
  e.red1 = red1;
  e.y_global = nullptr;
  e.z_global = nullptr;
  e.stats_global = nullptr;
  e.stage = nullptr;
  e.stage_ld = 0;
  e.head_out = a.out;
  e.head_b = a.head_b;
  e.rows = a.rows;
  e.has_ln = 1;
  e.act = 1;
  e.eps = a.eps;
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t row0 = tile * TM;
    e.row0 = row0;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    {   // the tile's positional encodings: 64 rows x 128 B, one 16-byte piece per thread.  (The previous tile's last
        // reads of this part of LDS -- rows of y1 -- lie in front of its epilogue's barriers.)
      const int r = tid_ >> 3, p = tid_ & 7;
      const int64_t row = row0 + r < a.rows ? row0 + r : a.rows - 1;
      *(u32x4*)(ps + r * kLd0 + p * 8) = *(const u32x4*)(a.pe + row * kK0 + p * 8);
    }
    __syncthreads();
    {
      f32x16 acc[2][2];
      gemm_rows64<2>(acc, ps, kLd0, kK0 / 16, a.w[0], tid_);
      e.gam_s = gam0;
      e.bet_s = bet0;
      e.head_s = nullptr;
      e.y_lds = ys;
      e.ld_lds = kLd1;
      e.drop = a.drop[0];
      e.z_global = STORE == 2 ? a.z_out[0] + tile * (TM * kN0) : a.z_out[0];
      e.stats_global = a.stats_out[0];
      e.stage = ys;
      e.stage_ld = kLd1;
      layer_epilogue<2, true, DROP, TRAIN, STORE>(acc, e);
    }
    __syncthreads();   // y0 complete
    if (a.y_out[0]) copy_tile_out<kN0>(ys, kLd1, a.y_out[0], row0, a.rows);
    {
      f32x16 acc[4][2];
      gemm_rows64<4>(acc, ys, kLd1, kN0 / 16, a.w[1], tid_);
      e.gam_s = gam1;
      e.bet_s = bet1;
      e.y_lds = ys;
      e.ld_lds = kLd2;
      e.drop = a.drop[1];
      e.z_global = STORE == 2 ? a.z_out[1] + tile * (TM * kN1) : a.z_out[1];
      e.stats_global = a.stats_out[1];
      e.stage_ld = kLd2;
      layer_epilogue<4, false, DROP, TRAIN, STORE>(acc, e);
    }
    __syncthreads();   // y1 complete
    if (a.y_out[1]) copy_tile_out<kN1>(ys, kLd2, a.y_out[1], row0, a.rows);
    {
      f32x16 acc[4][2];
      gemm_rows64<4>(acc, ys, kLd2, kN1 / 16, a.w[2], tid_);
      e.gam_s = gam2;
      e.bet_s = bet2;
      e.head_s = head_s;
      e.y_lds = STORE == 1 ? ys : nullptr;   // (training with saved activations for the operator chain: y2 leaves through the tile as well)
      e.drop = a.drop[2];
      e.z_global = STORE == 2 ? a.z_out[2] + tile * (TM * kN2) : a.z_out[2];
      e.stats_global = a.stats_out[2];
      layer_epilogue<4, false, DROP, TRAIN, STORE>(acc, e);   // ends with the head's barriers: every wave is past its reads of y1
      if (STORE == 1) {
        copy_tile_out<kN2>(ys, kLd2, a.y2_out, row0, a.rows);
        __syncthreads();   // (the next tile's positional encodings land inside this region)
      }
    }
  }
}

// ---- the decoder's BACKWARD pass with recompute, one launch (occ_base.py:99-153 under autograd).  The training forward
// above keeps nothing but the logits; this kernel walks a 64-row tile forward again (same numbers: z rounded to bf16 in
// front of every LayerNorm) and then backward, with the activations of the tile in LDS:
//   F0  pe -> z0 -> y0            (z0 parked, y0 -> tile and -> global: the weight-gradient operand of layer 1)
//   F1  y0 -> z1 -> y1            (z1 parked, y1 -> tile and -> global: the operand of layer 2)
//   F2  y1 -> z2 -> y2            (z2 parked; y2 only meets d logit: d head_w)
//   B2  d y2 = d logit x head_w;  LayerNorm / GELU / dropout backward -> d z2 (-> tile, -> global)
//   B1  d y1 = d z2 W2  (MFMA, W2^T fragments); LayerNorm backward with the parked z1 -> d z1 (-> tile, -> global)
//   B0  d y0 = d z1 W1;            LayerNorm backward with the parked z0 -> d z0 (-> global)
// "Parked": each lane stores the bf16 pairs of ITS accumulator elements to a per-workgroup strip of global memory and reads
// the same addresses back (twice: the LayerNorm backward needs xhat on both sides of the row sums) -- 5 KB per row, written
// and read by one CU, never by another.  What leaves for good is what the weight gradients contract over the rows:
// y0, y1, d z0, d z1, d z2 (8 KB per row; the forward used to leave 10 KB per row and the backward chain moved ~45 KB per
// row through HBM), and per workgroup the sums for d gamma, d beta of the three LayerNorms and d head_w.
// The LayerNorm backward runs on the accumulator layout of the MFMA (a lane holds 16 NPW channels of 2 rows): the row sums
// go through LDS like the forward's statistics, the per-channel sums over a tile's rows through 5 lane exchanges, and
// the lane that owns a channel adds them to the workgroup's strip of `partials` (one adder per address: fixed order).
constexpr int kBwdPartial = 2 * (kN0 + kN1 + kN2) + kN2;   // dg0 db0 dg1 db1 dg2 db2 dhead
constexpr int kBwdScratchWords = TM * (kN0 + kN1 + kN2) / 2;   // u32 words per workgroup: z0 | z1 | z2, bf16 pairs
constexpr int kOccBwdLds = TM * kLd2 * 2 + 2 * kWaves * TM * 4 + (2 * (kN0 + kN1 + kN2) + kN2) * 4 + 3 * TM * 2 * 4;
static_assert(kOccBwdLds <= 160 * 1024, "LDS budget");
struct OccMlpBwdArgs {
  const uint16_t* pe;        // [rows, 64] bf16
  const float* add;          // [*, 512] f32
  const int32_t* add_idx;    // [rows]
  const uint16_t* w[3];      // forward fragments ([512,64], [1024,512], [1024,1024])
  const uint16_t* wt[2];     // fragments of W1^T [512, 1024] and W2^T [1024, 1024]: the input-gradient operands
  const float* ln_w[3];
  const float* ln_b[3];
  const float* head_w;       // [1024]
  const float* dlogit;       // [rows] f32
  uint16_t* y_out[2];        // y0 [rows, 512], y1 [rows, 1024]
  uint16_t* dz_out[3];       // dz0 [rows, 512], dz1 [rows, 1024], dz2 [rows, 1024]
  float* partials;           // [gridDim.x][kBwdPartial], zero on entry
  uint32_t* scratch;         // RECOMPUTE: [gridDim.x][kBwdScratchWords]
  const uint32_t* z_in[3];   // no RECOMPUTE: what the forward parked (STORE 2), [tiles][64 N_l / 2] words per layer ...
  const float* stats_in[3];  // ... and its row statistics [rows, 2]
  float eps;
  int64_t rows;
  LnDropout drop[3];
};

struct BwdEpi {
  const float* gam_s;        // LDS [N]
  const float* bet_s;
  const float* head_s;       // LDS [N]: bf16-rounded head weight (layer 2 only)
  const float* add;
  const int32_t* add_idx;
  float* red0;
  float* red1;
  float* stat_s;             // LDS [64][2] of this layer: mean, rstd
  uint32_t* zpark;           // this workgroup's strip for this layer's z
  float* part_g;             // this workgroup's d gamma strip of this layer (d beta follows at + N)
  float* part_head;          // d head_w strip (layer 2)
  const float* dlogit;
  uint16_t* tile;            // LDS tile that receives y (forward) / dz (backward)
  int32_t ld;
  int64_t row0, rows;
  float eps;
  LnDropout drop;
};

__device__ __forceinline__ float bf16_round_f(float v) { return __uint_as_float(((uint32_t)ococc_f32_to_bf16(v)) << 16); }

// sum over the 32 rows of a lane group (lanes that share h), result valid in every lane: four steps on the VALU's data-
// parallel lane paths (quad swaps, mirrored halves: after each step the lanes that have been combined hold the same sum, so
// a mirror pairs what an exchange would), the last -- across the two rows of 16 lanes -- through the LDS crossbar
__device__ __forceinline__ float rows32_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));   // row_mirror
  v += __shfl_xor(v, 16, 64);
  return v;
}

// Forward half of a layer inside the backward kernel: as layer_epilogue<.., TRAIN> (same operations in the same order, so
// the same z, statistics and y as the training forward produced), but z is parked per lane and the statistics stay in
// LDS.  HEAD (the last layer): nothing else -- y2 is only needed where it meets d logit (d head_w), and the backward half
// of the layer computes it there from the parked z.
template <int NPW, bool ADD, bool MAY_DROP, bool HEAD>
__device__ __forceinline__ void recompute_epilogue(f32x16 (&acc)[NPW][2], const BwdEpi& e) {
  constexpr int N = NPW * 32 * kWaves;
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int m = tid_ & 31, h = (tid_ >> 5) & 1, wave = tid_ >> 6, nb0 = wave * NPW;
  int64_t rows_of[2];
  int32_t gidx[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    rows_of[mb] = e.row0 + 32 * mb + m;
    gidx[mb] = (ADD && rows_of[mb] < e.rows) ? e.add_idx[rows_of[mb]] : 0;
  }
  if (ADD) {
#pragma unroll
    for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = 32 * (nb0 + nb) + 8 * q + 4 * h;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const f32x4 g = *(const f32x4*)(e.add + (int64_t)gidx[mb] * N + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[nb][mb][4 * q + r] += g[r];
        }
        if (q == 3) __builtin_amdgcn_sched_barrier(0);
      }
  }
  // z rounded to bf16 (the LayerNorm works on the rounded values) and parked
#pragma unroll
  for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        u32x2 o;
        o.x = ln_pack2(ln_f32x2{acc[nb][mb][4 * q], acc[nb][mb][4 * q + 1]});
        o.y = ln_pack2(ln_f32x2{acc[nb][mb][4 * q + 2], acc[nb][mb][4 * q + 3]});
        acc[nb][mb][4 * q] = __uint_as_float(o.x << 16);
        acc[nb][mb][4 * q + 1] = __uint_as_float(o.x & 0xffff0000u);
        acc[nb][mb][4 * q + 2] = __uint_as_float(o.y << 16);
        acc[nb][mb][4 * q + 3] = __uint_as_float(o.y & 0xffff0000u);
        *(u32x2*)(e.zpark + ((((nb * 4 + q) * 2 + mb) * kThreads + tid_) << 1)) = o;
      }
  float rstd[2];
  {
    float sm[2] = {0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) sm[mb] += acc[nb][mb][i];
      sm[mb] += __shfl_xor(sm[mb], 32, 64);
      if (h == 0) e.red0[wave * TM + 32 * mb + m] = sm[mb];
    }
    __syncthreads();
    float q2[2], mean_[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) t += e.red0[w * TM + 32 * mb + m];
      const float mean = t / (float)N;
      mean_[mb] = mean;
      q2[mb] = 0.f;
#pragma unroll
      for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[nb][mb][i] -= mean;
          q2[mb] += acc[nb][mb][i] * acc[nb][mb][i];
        }
      q2[mb] += __shfl_xor(q2[mb], 32, 64);
      if (h == 0) e.red1[wave * TM + 32 * mb + m] = q2[mb];
    }
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) t += e.red1[w * TM + 32 * mb + m];
      rstd[mb] = rsqrtf(t / (float)N + e.eps);
      if (wave == 0 && h == 0) {
        e.stat_s[(32 * mb + m) * 2] = mean_[mb];
        e.stat_s[(32 * mb + m) * 2 + 1] = rstd[mb];
      }
    }
  }
  if (HEAD) return;
#pragma unroll
  for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = 32 * (nb0 + nb) + 8 * q + 4 * h;
      const f32x4 gm = *(const f32x4*)(e.gam_s + n), bt = *(const f32x4*)(e.bet_s + n);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        ln_f32x2 v0 = {acc[nb][mb][4 * q] * rstd[mb] * gm[0] + bt[0], acc[nb][mb][4 * q + 1] * rstd[mb] * gm[1] + bt[1]};
        ln_f32x2 v1 = {acc[nb][mb][4 * q + 2] * rstd[mb] * gm[2] + bt[2], acc[nb][mb][4 * q + 3] * rstd[mb] * gm[3] + bt[3]};
        v0 = ln_gelu2(v0);
        v1 = ln_gelu2(v1);
        if (MAY_DROP && e.drop.thr) {
          v0 = v0 * ln_dropout_mask2(e.drop, rows_of[mb], n >> 1, N >> 1);
          v1 = v1 * ln_dropout_mask2(e.drop, rows_of[mb], (n >> 1) + 1, N >> 1);
        }
        u32x2 o;
        o.x = ln_pack2(v0);
        o.y = ln_pack2(v1);
        *(u32x2*)(e.tile + (32 * mb + m) * e.ld + n) = o;
      }
      if (q & 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// Backward half of a layer: the accumulators hold d y (f32 sums of the input-gradient GEMM; GEN: nothing -- d y2 is
// bf16(d logit) x bf16(head_w)), rounded to bf16 as the operator chain hands it on; LayerNorm / GELU / dropout backward
// with the arithmetic of ln_bwd_piece8 / ln_bwd_finish8 (ln_math.hpp) per channel pair; d z -> the LDS tile (bf16).
//
// Shape of the code.  A lane owns 4-channel groups of two rows; in the row-major tile those are 8-byte slots nobody else
// touches.  So, behind a barrier (every wave is past its GEMM reads of the tile), each lane puts its d y groups into ITS
// slots, and two ROLLED loops over the groups follow: the first leaves only sums (the rows' two sums through LDS, the
// channels' d gamma / d beta through five lane exchanges and the owner lane's add to the workgroup's strip), the second
// computes d z and overwrites the d y slot.  Both read d y from LDS and the parked z from global memory, one group
// ahead.  (First version: everything unrolled on the accumulator registers, the first pass's 128 products kept for the
// second -- 6 000 instructions per instantiation, 400-700 spilled registers, and, the vector-memory counter retiring in
// order, every reload waiting for the atomics and stores in front of it: 64 % of the epilogue's cycles in s_waitcnt with
// the VALU 13 % busy, tools/pmc_decoder_bwd.sh; 4.4 ms per 1 M rows and 1024 channels against 1.5 ms for the stand-alone
// LayerNorm-backward kernel.)
template <int NPW, bool MAY_DROP, bool GEN>
__device__ __forceinline__ void lnbwd_epilogue(f32x16 (&acc)[NPW][2], const BwdEpi& e) {
  constexpr int N = NPW * 32 * kWaves;
  constexpr int G = NPW * 4;   // 4-channel groups per lane and row
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int m = tid_ & 31, h = (tid_ >> 5) & 1, wave = tid_ >> 6, nb0 = wave * NPW;
  if (!GEN) {
    u32x2 dyp[G * 2];
#pragma unroll
    for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          u32x2 o;
          o.x = ln_pack2(ln_f32x2{acc[nb][mb][4 * q], acc[nb][mb][4 * q + 1]});
          o.y = ln_pack2(ln_f32x2{acc[nb][mb][4 * q + 2], acc[nb][mb][4 * q + 3]});
          // rows past the end: clamped copies of the last row, no gradient
          if (e.row0 + 32 * mb + m >= e.rows) o = u32x2{0u, 0u};
          dyp[(nb * 4 + q) * 2 + mb] = o;
        }
    __syncthreads();   // the tile is free: every wave has read its last operands
#pragma unroll
    for (int nb = 0; nb < NPW; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
          *(u32x2*)(e.tile + (32 * mb + m) * e.ld + 32 * (nb0 + nb) + 8 * q + 4 * h) = dyp[(nb * 4 + q) * 2 + mb];
  }
  int64_t rows_of[2];
  float mean[2], rstd[2], dlb[2], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    rows_of[mb] = e.row0 + 32 * mb + m;
    mean[mb] = e.stat_s[(32 * mb + m) * 2];
    rstd[mb] = e.stat_s[(32 * mb + m) * 2 + 1];
    dlb[mb] = (GEN && rows_of[mb] < e.rows) ? bf16_round_f(e.dlogit[rows_of[mb]]) : 0.f;
  }
  const uint32_t* zlane = e.zpark + (tid_ << 1);           // + (g * 2 + mb) * 2 kThreads words
  uint16_t* slot0 = e.tile + m * e.ld + 32 * nb0 + 4 * h;   // + 32 mb ld + 8 g elements  (32 nb + 8 q = 8 g)
  const int n0 = 32 * nb0 + 4 * h;
  // d (LayerNorm output) of one group and its xhat: what both passes start from
  auto group = [&](int g, int mb, const u32x2 zp, const f32x4& gm, const f32x4& bt, const f32x4& hw, ln_f32x2 (&xh)[2],
                   ln_f32x2 (&d)[2], bool want_y, ln_f32x2 (&yv)[2]) {
    const int n = n0 + 8 * g;
    xh[0] = ln_f32x2{__uint_as_float(zp.x << 16), __uint_as_float(zp.x & 0xffff0000u)};
    xh[1] = ln_f32x2{__uint_as_float(zp.y << 16), __uint_as_float(zp.y & 0xffff0000u)};
    if (GEN) {
      d[0] = ln_f32x2{bf16_round_f(dlb[mb] * hw[0]), bf16_round_f(dlb[mb] * hw[1])};
      d[1] = ln_f32x2{bf16_round_f(dlb[mb] * hw[2]), bf16_round_f(dlb[mb] * hw[3])};
    } else {
      const u32x2 dp = *(const u32x2*)(slot0 + 32 * mb * e.ld + 8 * g);
      d[0] = ln_f32x2{__uint_as_float(dp.x << 16), __uint_as_float(dp.x & 0xffff0000u)};
      d[1] = ln_f32x2{__uint_as_float(dp.y << 16), __uint_as_float(dp.y & 0xffff0000u)};
    }
    ln_f32x2 k0 = {1.f, 1.f}, k1 = {1.f, 1.f};
    if (MAY_DROP && e.drop.thr) {
      k0 = ln_dropout_mask2(e.drop, rows_of[mb], n >> 1, N >> 1);
      k1 = ln_dropout_mask2(e.drop, rows_of[mb], (n >> 1) + 1, N >> 1);
      d[0] = d[0] * k0;
      d[1] = d[1] * k1;
    }
    xh[0] = (xh[0] - mean[mb]) * rstd[mb];
    xh[1] = (xh[1] - mean[mb]) * rstd[mb];
    const ln_f32x2 t0 = xh[0] * ln_f32x2{gm[0], gm[1]} + ln_f32x2{bt[0], bt[1]};
    const ln_f32x2 t1 = xh[1] * ln_f32x2{gm[2], gm[3]} + ln_f32x2{bt[2], bt[3]};
    if (want_y) {   // y2 as the forward computed it (same expression, same rounding): it meets d logit in d head_w
      const uint32_t p0 = ln_pack2(ln_gelu2(t0) * k0), p1 = ln_pack2(ln_gelu2(t1) * k1);
      yv[0] = ln_f32x2{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u)};
      yv[1] = ln_f32x2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
    }
    d[0] = d[0] * ln_gelu_grad2(t0);
    d[1] = d[1] * ln_gelu_grad2(t1);
  };
  auto load_z = [&](int g, u32x2 (&z)[2]) {
    z[0] = *(const u32x2*)(zlane + (size_t)(g * 2) * (2 * kThreads));
    z[1] = *(const u32x2*)(zlane + (size_t)(g * 2 + 1) * (2 * kThreads));
  };
  // ---- first pass: the sums
  {
    u32x2 zc[2], zn[2];
    load_z(0, zc);
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
      load_z(g + 1 < G ? g + 1 : g, zn);
      const int n = n0 + 8 * g;
      const f32x4 gm = *(const f32x4*)(e.gam_s + n), bt = *(const f32x4*)(e.bet_s + n);
      f32x4 hw = {0.f, 0.f, 0.f, 0.f};
      if (GEN) hw = *(const f32x4*)(e.head_s + n);
      ln_f32x2 dg0 = {0.f, 0.f}, dg1 = {0.f, 0.f}, db0 = {0.f, 0.f}, db1 = {0.f, 0.f}, dh0 = {0.f, 0.f}, dh1 = {0.f, 0.f};
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        ln_f32x2 xh[2], d[2], yv[2];
        group(g, mb, zc[mb], gm, bt, hw, xh, d, GEN, yv);
        if (GEN) {
          dh0 += yv[0] * dlb[mb];
          dh1 += yv[1] * dlb[mb];
        }
        dg0 += d[0] * xh[0];
        dg1 += d[1] * xh[1];
        db0 += d[0];
        db1 += d[1];
        const ln_f32x2 z0 = d[0] * ln_f32x2{gm[0], gm[1]}, z1 = d[1] * ln_f32x2{gm[2], gm[3]};
        const ln_f32x2 a1 = z0 + z1, a2 = z0 * xh[0] + z1 * xh[1];
        s1[mb] += a1.x + a1.y;
        s2[mb] += a2.x + a2.y;
      }
      float v[8] = {dg0.x, dg0.y, dg1.x, dg1.y, db0.x, db0.y, db1.x, db1.y};
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = rows32_sum(v[r]);
      if (m == 0) {   // (one adder per address: fixed order)
#pragma unroll
        for (int r = 0; r < 8; ++r) unsafeAtomicAdd(e.part_g + (r >> 2) * N + n + (r & 3), v[r]);
      }
      if (GEN) {   // d head_w[n] = sum over the rows of bf16(d logit) x y2 (bf16), as the library GEMM on those operands
        float u[4] = {dh0.x, dh0.y, dh1.x, dh1.y};
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = rows32_sum(u[r]);
        if (m == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) unsafeAtomicAdd(e.part_head + n + r, u[r]);
        }
      }
      zc[0] = zn[0];
      zc[1] = zn[1];
    }
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    s1[mb] += __shfl_xor(s1[mb], 32, 64);
    s2[mb] += __shfl_xor(s2[mb], 32, 64);
    if (h == 0) {
      e.red0[wave * TM + 32 * mb + m] = s1[mb];
      e.red1[wave * TM + 32 * mb + m] = s2[mb];
    }
  }
  __syncthreads();
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      t1 += e.red0[w * TM + 32 * mb + m];
      t2 += e.red1[w * TM + 32 * mb + m];
    }
    s1[mb] = t1 * (1.f / N);
    s2[mb] = t2 * (1.f / N);
  }
  // ---- second pass: d z = ((dz gamma - s1) - xhat s2) rstd -> the lane's slot of the tile
  {
    u32x2 zc[2], zn[2];
    load_z(0, zc);
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
      load_z(g + 1 < G ? g + 1 : g, zn);
      const int n = n0 + 8 * g;
      const f32x4 gm = *(const f32x4*)(e.gam_s + n), bt = *(const f32x4*)(e.bet_s + n);
      f32x4 hw = {0.f, 0.f, 0.f, 0.f};
      if (GEN) hw = *(const f32x4*)(e.head_s + n);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        ln_f32x2 xh[2], d[2], yv[2];
        group(g, mb, zc[mb], gm, bt, hw, xh, d, false, yv);
        const ln_f32x2 z0 = d[0] * ln_f32x2{gm[0], gm[1]}, z1 = d[1] * ln_f32x2{gm[2], gm[3]};
        u32x2 o;
        o.x = ln_pack2(((z0 - s1[mb]) - xh[0] * s2[mb]) * rstd[mb]);
        o.y = ln_pack2(((z1 - s1[mb]) - xh[1] * s2[mb]) * rstd[mb]);
        *(u32x2*)(slot0 + 32 * mb * e.ld + 8 * g) = o;
      }
      zc[0] = zn[0];
      zc[1] = zn[1];
    }
  }
  __syncthreads();   // d z complete in the tile
}

#ifndef OCOCC_BWD_PHASES
#define OCOCC_BWD_PHASES 31   // diagnostic builds (tools/probe/decoder_bwd_phases.sh): 1 F0 + F1, 2 F2, 4 B2, 8 B1, 16 B0
#endif
template <bool DROP, bool RECOMPUTE>
__global__ void __launch_bounds__(kThreads, 2)
occ_mlp_bwd_kernel(OccMlpBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* ys = (uint16_t*)smem;                 // the tile: y0 / y1 / dz2 / dz1 / dz0 in turn
  uint16_t* ps = ys + kPeOffset;                  // positional-encoding tile [64][kLd0]
  float* red0 = (float*)(ys + TM * kLd2);
  float* red1 = red0 + kWaves * TM;
  float* gam0 = red1 + kWaves * TM;
  float* bet0 = gam0 + kN0;
  float* gam1 = bet0 + kN0;
  float* bet1 = gam1 + kN1;
  float* gam2 = bet1 + kN1;
  float* bet2 = gam2 + kN2;
  float* head_s = bet2 + kN2;
  float* stat_s = head_s + kN2;                   // [3][64][2]
  for (int i = threadIdx.x; i < kN1; i += kThreads) {
    if (i < kN0) {
      gam0[i] = a.ln_w[0][i];
      bet0[i] = a.ln_b[0][i];
    }
    gam1[i] = a.ln_w[1][i];
    bet1[i] = a.ln_b[1][i];
    gam2[i] = a.ln_w[2][i];
    bet2[i] = a.ln_b[2][i];
    head_s[i] = bf16_round_f(a.head_w[i]);
  }
  const int64_t tiles = (a.rows + TM - 1) / TM;
  uint32_t* park = a.scratch + (int64_t)blockIdx.x * kBwdScratchWords;
  float* part = a.partials + (int64_t)blockIdx.x * kBwdPartial;
  BwdEpi e;
  e.add = a.add;
  e.add_idx = a.add_idx;
  e.red0 = red0;
  e.red1 = red1;
  e.head_s = head_s;
  e.part_head = part + 2 * (kN0 + kN1 + kN2);
  e.dlogit = a.dlogit;
  e.tile = ys;
  e.rows = a.rows;
  e.eps = a.eps;
  auto layer = [&](int l) {
    e.gam_s = l == 0 ? gam0 : (l == 1 ? gam1 : gam2);
    e.bet_s = l == 0 ? bet0 : (l == 1 ? bet1 : bet2);
    e.stat_s = stat_s + l * TM * 2;
    if (RECOMPUTE) e.zpark = park + (l == 0 ? 0 : (l == 1 ? TM * kN0 / 2 : TM * (kN0 + kN1) / 2));
    else e.zpark = const_cast<uint32_t*>(a.z_in[l]) + (e.row0 / TM) * (TM * (l == 0 ? kN0 : kN1) / 2);
    e.part_g = part + (l == 0 ? 0 : (l == 1 ? 2 * kN0 : 2 * (kN0 + kN1)));
    e.drop = a.drop[l];
  };
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t row0 = tile * TM;
    e.row0 = row0;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    if (RECOMPUTE) {
      {
        const int r = tid_ >> 3, p = tid_ & 7;
        const int64_t row = row0 + r < a.rows ? row0 + r : a.rows - 1;
        *(u32x4*)(ps + r * kLd0 + p * 8) = *(const u32x4*)(a.pe + row * kK0 + p * 8);
      }
      __syncthreads();
      if (OCOCC_BWD_PHASES & 1) {   // F0
        f32x16 acc[2][2];
        gemm_rows64<2>(acc, ps, kLd0, kK0 / 16, a.w[0], tid_);
        layer(0);
        e.ld = kLd1;
        recompute_epilogue<2, true, DROP, false>(acc, e);
      }
      __syncthreads();   // y0 complete
      copy_tile_out<kN0>(ys, kLd1, a.y_out[0], row0, a.rows);
      if (OCOCC_BWD_PHASES & 1) {   // F1
        f32x16 acc[4][2];
        gemm_rows64<4>(acc, ys, kLd1, kN0 / 16, a.w[1], tid_);
        layer(1);
        e.ld = kLd2;
        recompute_epilogue<4, false, DROP, false>(acc, e);
      }
      __syncthreads();   // y1 complete
      copy_tile_out<kN1>(ys, kLd2, a.y_out[1], row0, a.rows);
    } else if (tid_ < 3 * TM * 2) {
      // the statistics the forward left: [layer][row][mean | rstd] -> LDS (the first barrier inside B2 makes them visible)
      const int l = tid_ / (TM * 2), r = (tid_ % (TM * 2)) >> 1, c = tid_ & 1;
      const int64_t row = row0 + r < a.rows ? row0 + r : a.rows - 1;
      stat_s[tid_] = a.stats_in[l][row * 2 + c];
    }
    {   // F2 + B2
      f32x16 acc[4][2];
      layer(2);
      e.ld = kLd2;
      if (RECOMPUTE && (OCOCC_BWD_PHASES & 2)) {
        gemm_rows64<4>(acc, ys, kLd2, kN1 / 16, a.w[2], tid_);
        recompute_epilogue<4, false, DROP, true>(acc, e);
      }
      __syncthreads();   // (the statistics' partial sums in red0 / red1 have been read; z2 is parked, the statistics are in LDS)
      if (OCOCC_BWD_PHASES & 4) lnbwd_epilogue<4, DROP, true>(acc, e);
    }
    copy_tile_out<kN2>(ys, kLd2, a.dz_out[2], row0, a.rows);
    if (OCOCC_BWD_PHASES & 8) {   // B1: d y1 = d z2 W2
      f32x16 acc[4][2];
      gemm_rows64<4>(acc, ys, kLd2, kN2 / 16, a.wt[1], tid_);
      layer(1);
      e.ld = kLd2;
      lnbwd_epilogue<4, DROP, false>(acc, e);
    }
    copy_tile_out<kN1>(ys, kLd2, a.dz_out[1], row0, a.rows);
    if (OCOCC_BWD_PHASES & 16) {   // B0: d y0 = d z1 W1
      f32x16 acc[2][2];
      gemm_rows64<2>(acc, ys, kLd2, kN1 / 16, a.wt[0], tid_);
      layer(0);
      e.ld = kLd1;
      lnbwd_epilogue<2, DROP, false>(acc, e);
    }
    copy_tile_out<kN0>(ys, kLd1, a.dz_out[0], row0, a.rows);
    __syncthreads();   // (the next tile's positional encodings land inside the tile)
  }
}

// f32 matrices (any strides) -> bf16 A-operand fragments of v_mfma_f32_32x32x16_bf16:
//   dst[rb][cs][lane][j] = S[32 rb + (lane & 31)][16 cs + 8 (lane >> 5) + j]
constexpr int kMaxFrag = 16;
struct Frag32Pack {
  const float* src[kMaxFrag];
  uint16_t* dst[kMaxFrag];
  int32_t rows[kMaxFrag], cols[kMaxFrag], true_cols[kMaxFrag];
  int64_t rs[kMaxFrag], cs[kMaxFrag];
  int32_t first_block[kMaxFrag + 1];
  int32_t count;
};
__global__ void __launch_bounds__(256) linear_fragments32_kernel(Frag32Pack pk) {
  int t = 0;
  while (t + 1 < pk.count && (int)blockIdx.x >= pk.first_block[t + 1]) ++t;
  const int cols = pk.cols[t], total = pk.rows[t] * cols, ksteps = cols >> 4;
  const int nblk = pk.first_block[t + 1] - pk.first_block[t];
  for (int i = ((int)blockIdx.x - pk.first_block[t]) * 256 + (int)threadIdx.x; i < total; i += nblk * 256) {
    const int j = i & 7, lane = (i >> 3) & 63, blk = i >> 9;
    const int cs = blk % ksteps, rb = blk / ksteps;
    const int r = 32 * rb + (lane & 31), c = 16 * cs + 8 * (lane >> 5) + j;
    pk.dst[t][i] = c < pk.true_cols[t] ? ococc_f32_to_bf16(pk.src[t][r * pk.rs[t] + c * pk.cs[t]]) : (uint16_t)0;
  }
}

// A11: positional encoding of the query points as the first layer's bf16 operand (PosEncode.forward,
// mmdet3d/models/occ/occ_base.py:33-57: x^ = (x - lo) / (hi - lo) * 2 - 1, then sin | cos of pi 2^l x^, laid out [2L][3]),
// columns 6L .. ld - 1 zero.  A thread writes 8 columns of a row.  Same operation order and the same sinf / cosf as the
// torch expression (f32), so the bf16 roundings agree.
struct PosEncodeArgs {
  const float* xyz;
  uint16_t* out;
  int64_t rows;
  float lo[3], span[3];
  int32_t L, ld, use_norm;
};
__global__ void __launch_bounds__(256)
pos_encode_kernel(PosEncodeArgs a) {
  const int groups = a.ld >> 3;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.rows * groups) return;
  const int64_t row = i / groups;
  const int c0 = (int)(i - row * groups) * 8;
  float p[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float x = a.xyz[row * 3 + d];
    if (a.use_norm) x = (x - a.lo[d]) / a.span[d] * 2.0f - 1.0f;
    p[d] = x;
  }
  const float pi = 3.14159265358979323846f;
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    uint16_t v = 0;
    if (c < 6 * a.L) {
      const int cos_half = c >= 3 * a.L;
      const int cc = cos_half ? c - 3 * a.L : c;
      const int l = cc / 3, d = cc - 3 * l;
      const float arg = pi * (p[d] * (float)(1 << l));
      v = ococc_f32_to_bf16(cos_half ? cosf(arg) : sinf(arg));
    }
    if (j & 1) w[j >> 1] |= (uint32_t)v << 16; else w[j >> 1] = v;
  }
  *(u32x4*)(a.out + row * a.ld + c0) = u32x4{w[0], w[1], w[2], w[3]};
}

int cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  return cus;
}

}  // namespace

extern "C" int ococc_linear_fragments32_bf16(int32_t count, const void* const* src, const int64_t* rows,
                                             const int64_t* cols, const int64_t* padded_cols, const int64_t* row_stride,
                                             const int64_t* col_stride, void* const* dst, ococc_stream_t stream) {
  OCOCC_REQUIRE(count >= 0 && count <= kMaxFrag, "at most 16 matrices per call");
  if (count == 0) return OCOCC_OK;
  OCOCC_REQUIRE(src && rows && cols && padded_cols && row_stride && col_stride && dst, "null pointer table");
  Frag32Pack pk;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    OCOCC_REQUIRE(src[i] && dst[i] && rows[i] > 0 && cols[i] > 0 && rows[i] % 32 == 0 && padded_cols[i] % 16 == 0 &&
                      padded_cols[i] >= cols[i] && rows[i] * padded_cols[i] < (1 << 30),
                  "a matrix needs rows in multiples of 32 and padded columns in multiples of 16");
    pk.src[i] = (const float*)src[i];
    pk.dst[i] = (uint16_t*)dst[i];
    pk.rows[i] = (int32_t)rows[i];
    pk.cols[i] = (int32_t)padded_cols[i];
    pk.true_cols[i] = (int32_t)cols[i];
    pk.rs[i] = row_stride[i];
    pk.cs[i] = col_stride[i];
    pk.first_block[i] = blocks;
    blocks += (int)(ococc_cdiv(rows[i] * padded_cols[i], 256) < 128 ? ococc_cdiv(rows[i] * padded_cols[i], 256) : 128);
  }
  pk.first_block[count] = blocks;
  pk.count = count;
  hipLaunchKernelGGL(linear_fragments32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pk);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_mlp_layer_fwd_bf16(const uint16_t* x, int64_t rows, int32_t k, const uint16_t* w_frag, int32_t n,
                                        const float* bias, const float* add_rows, const int32_t* add_index,
                                        const float* ln_weight, const float* ln_bias, float eps, int32_t act,
                                        uint32_t drop_threshold, uint64_t dropout_seed, uint16_t* y,
                                        const float* head_weight, const float* head_bias, float* head_out,
                                        ococc_stream_t stream) {
  OCOCC_REQUIRE(rows >= 0 && (n == 512 || n == 1024) && k >= 64 && k <= 1024 && k % 64 == 0,
                "output width 512 or 1024, input width a multiple of 64 up to 1024");
  OCOCC_REQUIRE((act == 0 || act == 1) && (!ln_weight == !ln_bias) && (!add_rows == !add_index), "bad arguments");
  OCOCC_REQUIRE(drop_threshold < 65536u, "bad arguments");
  if (rows == 0) return OCOCC_OK;
  OCOCC_REQUIRE(x && w_frag && (!head_weight || head_out) && (y || head_weight), "null pointer");
  OCOCC_REQUIRE((((uintptr_t)x | (uintptr_t)w_frag | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)add_rows |
                  (uintptr_t)ln_weight | (uintptr_t)ln_bias | (uintptr_t)head_weight) & 15) == 0,
                "buffers must be 16-byte aligned");
  MlpLayerArgs a;
  a.x = x;
  a.add = add_rows;
  a.add_idx = add_index;
  a.wfrag = w_frag;
  a.ln_w = ln_weight;
  a.ln_b = ln_bias;
  a.bias = bias;
  a.y = y;
  a.head_w = head_weight;
  a.head_out = head_out;
  a.head_b = head_bias;
  a.eps = eps;
  a.act = act;
  a.K = k;
  a.N = n;
  a.rows = rows;
  const uint32_t thr = drop_threshold;
  a.drop.thr = thr;
  a.drop.scale = thr ? 65536.f / (65536.f - (float)thr) : 1.f;
  a.drop.seed_lo = (uint32_t)dropout_seed;
  a.drop.seed_hi = (uint32_t)(dropout_seed >> 32);
  const int lds = TM * (k + 16) * 2 + 2 * kWaves * TM * 4 + 4 * n * 4;
  const int64_t tiles = ococc_cdiv(rows, TM);
  const unsigned grid = (unsigned)(tiles < cu_count() ? tiles : cu_count());
#define OCOCC_LAUNCH(NPW_, ADD_)                                                                                          \
  do {                                                                                                                     \
    OCOCC_HIP(hipFuncSetAttribute((const void*)mlp_layer_fwd_kernel<NPW_, ADD_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  lds));                                                                                   \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(mlp_layer_fwd_kernel<NPW_, ADD_>), dim3(grid), dim3(kThreads), lds,                   \
                       (hipStream_t)stream, a);                                                                            \
  } while (0)
  const bool add = add_rows != nullptr;
  if (n == 512) {
    if (add) OCOCC_LAUNCH(2, true); else OCOCC_LAUNCH(2, false);
  } else {
    if (add) OCOCC_LAUNCH(4, true); else OCOCC_LAUNCH(4, false);
  }
#undef OCOCC_LAUNCH
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_pos_encode_bf16(const float* xyz, int64_t rows, const float* bound, int32_t num_freqs, uint16_t* out,
                                     int32_t ld, ococc_stream_t stream) {
  OCOCC_REQUIRE(rows >= 0 && num_freqs >= 1 && num_freqs <= 16 && ld % 8 == 0 && ld >= 6 * num_freqs,
                "1..16 frequencies, row stride a multiple of 8 holding 6 L columns");
  if (rows == 0) return OCOCC_OK;
  OCOCC_REQUIRE(xyz && out && ((uintptr_t)out & 15) == 0, "null or misaligned pointer");
  PosEncodeArgs a;
  a.xyz = xyz;
  a.out = out;
  a.rows = rows;
  a.L = num_freqs;
  a.ld = ld;
  a.use_norm = bound != nullptr;
  for (int d = 0; d < 3; ++d) {
    a.lo[d] = bound ? bound[d] : 0.f;
    a.span[d] = bound ? bound[3 + d] - bound[d] : 1.f;
  }
  const int64_t work = rows * (ld / 8);
  hipLaunchKernelGGL(pos_encode_kernel, dim3((unsigned)ococc_cdiv(work, 256)), dim3(256), 0, (hipStream_t)stream, a);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_occ_mlp_fwd_bf16(const uint16_t* pe, int64_t rows, const float* add_rows, const int32_t* add_index,
                                      const void* const* w_frag, const void* const* ln_weight, const void* const* ln_bias,
                                      float eps, const float* head_weight, const float* head_bias, uint32_t drop_threshold,
                                      const uint64_t* dropout_seeds, uint16_t* y0_out, uint16_t* y1_out, float* out,
                                      ococc_stream_t stream) {
  OCOCC_REQUIRE(rows >= 0 && drop_threshold < 65536u && (drop_threshold == 0 || dropout_seeds), "bad arguments");
  if (rows == 0) return OCOCC_OK;
  OCOCC_REQUIRE(pe && add_rows && add_index && w_frag && ln_weight && ln_bias && head_weight && out, "null pointer");
  OccMlpArgs a;
  uintptr_t align = (uintptr_t)pe | (uintptr_t)add_rows | (uintptr_t)head_weight | (uintptr_t)y0_out | (uintptr_t)y1_out;
  for (int l = 0; l < 3; ++l) {
    OCOCC_REQUIRE(w_frag[l] && ln_weight[l] && ln_bias[l], "null pointer");
    a.w[l] = (const uint16_t*)w_frag[l];
    a.ln_w[l] = (const float*)ln_weight[l];
    a.ln_b[l] = (const float*)ln_bias[l];
    align |= (uintptr_t)w_frag[l] | (uintptr_t)ln_weight[l] | (uintptr_t)ln_bias[l];
    a.drop[l].thr = drop_threshold;
    a.drop[l].scale = drop_threshold ? 65536.f / (65536.f - (float)drop_threshold) : 1.f;
    a.drop[l].seed_lo = drop_threshold ? (uint32_t)dropout_seeds[l] : 0u;
    a.drop[l].seed_hi = drop_threshold ? (uint32_t)(dropout_seeds[l] >> 32) : 0u;
  }
  OCOCC_REQUIRE((align & 15) == 0, "pointers must be 16-byte aligned");
  a.pe = pe;
  a.add = add_rows;
  a.add_idx = add_index;
  a.head_w = head_weight;
  a.head_b = head_bias;
  a.out = out;
  a.y_out[0] = y0_out;
  a.y_out[1] = y1_out;
  a.y2_out = nullptr;
  for (int l = 0; l < 3; ++l) {
    a.z_out[l] = nullptr;
    a.stats_out[l] = nullptr;
  }
  a.eps = eps;
  a.rows = rows;
  const int64_t tiles = ococc_cdiv(rows, TM);
  const unsigned grid = (unsigned)(tiles < cu_count() ? tiles : cu_count());
  if (drop_threshold) {
    OCOCC_HIP(hipFuncSetAttribute((const void*)occ_mlp_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kOccMlpLds));
    hipLaunchKernelGGL(occ_mlp_fwd_kernel<true>, dim3(grid), dim3(kThreads), kOccMlpLds, (hipStream_t)stream, a);
  } else {
    OCOCC_HIP(hipFuncSetAttribute((const void*)occ_mlp_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kOccMlpLds));
    hipLaunchKernelGGL(occ_mlp_fwd_kernel<false>, dim3(grid), dim3(kThreads), kOccMlpLds, (hipStream_t)stream, a);
  }
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

// The same launch for a training step: besides the logits it leaves what the backward pass reads -- per layer the
// LayerNorm input z (bf16), its row statistics and the activation y (bf16, after GELU and dropout).
extern "C" int ococc_occ_mlp_train_fwd_bf16(const uint16_t* pe, int64_t rows, const float* add_rows, const int32_t* add_index,
                                            const void* const* w_frag, const void* const* ln_weight,
                                            const void* const* ln_bias, float eps, const float* head_weight,
                                            const float* head_bias, uint32_t drop_threshold, const uint64_t* dropout_seeds,
                                            void* const* z_out, void* const* y_out, void* const* stats_out, float* out,
                                            ococc_stream_t stream) {
  OCOCC_REQUIRE(rows >= 0 && drop_threshold < 65536u && (drop_threshold == 0 || dropout_seeds), "bad arguments");
  if (rows == 0) return OCOCC_OK;
  OCOCC_REQUIRE(pe && add_rows && add_index && w_frag && ln_weight && ln_bias && head_weight && out, "null pointer");
  // z_out, y_out, stats_out all null: the forward of the recomputing backward (ococc_occ_mlp_bwd_bf16) -- the same
  // numbers, nothing kept but the logits
  const bool store = z_out != nullptr;
  OCOCC_REQUIRE((y_out != nullptr) == store && (stats_out != nullptr) == store, "z_out, y_out, stats_out: all or none");
  // y_out[2] null: z_out[l] are PARKED (the lane layout ococc_occ_mlp_bwd_bf16 reads without recompute: bf16
  // [ceil(rows / 64) * 64, n_l] elements each), y2 is not kept
  const bool parked = store && y_out[2] == nullptr;
  OccMlpArgs a;
  uintptr_t align = (uintptr_t)pe | (uintptr_t)add_rows | (uintptr_t)head_weight;
  for (int l = 0; l < 3; ++l) {
    OCOCC_REQUIRE(w_frag[l] && ln_weight[l] && ln_bias[l] && (!store || (z_out[l] && (y_out[l] || (parked && l == 2)) && stats_out[l])),
                  "null pointer");
    a.w[l] = (const uint16_t*)w_frag[l];
    a.ln_w[l] = (const float*)ln_weight[l];
    a.ln_b[l] = (const float*)ln_bias[l];
    a.z_out[l] = store ? (uint16_t*)z_out[l] : nullptr;
    a.stats_out[l] = store ? (float*)stats_out[l] : nullptr;
    align |= (uintptr_t)w_frag[l] | (uintptr_t)ln_weight[l] | (uintptr_t)ln_bias[l];
    if (store) align |= (uintptr_t)z_out[l] | (uintptr_t)(y_out[l] ? y_out[l] : nullptr);
    a.drop[l].thr = drop_threshold;
    a.drop[l].scale = drop_threshold ? 65536.f / (65536.f - (float)drop_threshold) : 1.f;
    a.drop[l].seed_lo = drop_threshold ? (uint32_t)dropout_seeds[l] : 0u;
    a.drop[l].seed_hi = drop_threshold ? (uint32_t)(dropout_seeds[l] >> 32) : 0u;
  }
  OCOCC_REQUIRE((align & 15) == 0, "pointers must be 16-byte aligned");
  a.pe = pe;
  a.add = add_rows;
  a.add_idx = add_index;
  a.head_w = head_weight;
  a.head_b = head_bias;
  a.out = out;
  a.y_out[0] = store ? (uint16_t*)y_out[0] : nullptr;
  a.y_out[1] = store ? (uint16_t*)y_out[1] : nullptr;
  a.y2_out = store ? (uint16_t*)y_out[2] : nullptr;
  a.eps = eps;
  a.rows = rows;
  const int64_t tiles = ococc_cdiv(rows, TM);
  const unsigned grid = (unsigned)(tiles < cu_count() ? tiles : cu_count());
#define OCOCC_LAUNCH_TRAIN(DROP_, STORE_)                                                                                \
  do {                                                                                                                    \
    OCOCC_HIP(hipFuncSetAttribute((const void*)occ_mlp_fwd_kernel<DROP_, true, STORE_>,                                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kOccMlpLds));                               \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(occ_mlp_fwd_kernel<DROP_, true, STORE_>), dim3(grid), dim3(kThreads), kOccMlpLds,   \
                       (hipStream_t)stream, a);                                                                           \
  } while (0)
  if (drop_threshold) {
    if (parked) OCOCC_LAUNCH_TRAIN(true, 2); else if (store) OCOCC_LAUNCH_TRAIN(true, 1); else OCOCC_LAUNCH_TRAIN(true, 0);
  } else {
    if (parked) OCOCC_LAUNCH_TRAIN(false, 2); else if (store) OCOCC_LAUNCH_TRAIN(false, 1); else OCOCC_LAUNCH_TRAIN(false, 0);
  }
#undef OCOCC_LAUNCH_TRAIN
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

// The backward pass of that forward with recompute (see occ_mlp_bwd_kernel).  dlogit f32 [rows]; w_t_frag: fragments of
// W1^T [512, 1024] and W2^T [1024, 1024]; leaves y0, y1, dz0, dz1, dz2 (bf16) for the weight gradients, and per workgroup
// the sums [dg0 | db0 | dg1 | db1 | dg2 | db2 | d head_w] in partials [ococc_occ_mlp_bwd_workgroups()][..._partial_cols()]
// (f32, ZERO on entry); scratch: ococc_occ_mlp_bwd_scratch_bytes().
extern "C" int64_t ococc_occ_mlp_bwd_workgroups(int64_t rows) {
  if (rows < 0) return -1;
  const int64_t tiles = ococc_cdiv(rows > 0 ? rows : 1, TM);
  return tiles < cu_count() ? tiles : cu_count();
}
extern "C" int64_t ococc_occ_mlp_bwd_partial_cols(void) { return kBwdPartial; }
extern "C" int64_t ococc_occ_mlp_bwd_scratch_bytes(int64_t rows) {
  const int64_t wgs = ococc_occ_mlp_bwd_workgroups(rows);
  return wgs < 0 ? -1 : wgs * (int64_t)kBwdScratchWords * 4;
}
extern "C" int ococc_occ_mlp_bwd_bf16(const uint16_t* pe, int64_t rows, const float* add_rows, const int32_t* add_index,
                                      const void* const* w_frag, const void* const* w_t_frag, const void* const* ln_weight,
                                      const void* const* ln_bias, float eps, const float* head_weight, const float* dlogit,
                                      uint32_t drop_threshold, const uint64_t* dropout_seeds, const void* const* z_parked,
                                      const void* const* stats, void* const* y_out, void* const* dz_out, float* partials,
                                      void* scratch, int64_t scratch_bytes, ococc_stream_t stream) {
  OCOCC_REQUIRE(rows >= 0 && drop_threshold < 65536u && (drop_threshold == 0 || dropout_seeds), "bad arguments");
  if (rows == 0) return OCOCC_OK;
  const bool recompute = z_parked == nullptr;
  OCOCC_REQUIRE((stats == nullptr) == recompute, "z_parked and stats go together");
  OCOCC_REQUIRE(w_t_frag && ln_weight && ln_bias && head_weight && dlogit && dz_out && partials, "null pointer");
  OCOCC_REQUIRE(!recompute || (pe && add_rows && add_index && w_frag && y_out && scratch), "null pointer (recompute)");
  OCOCC_REQUIRE(!recompute || scratch_bytes >= ococc_occ_mlp_bwd_scratch_bytes(rows), "scratch too small");
  OccMlpBwdArgs a;
  uintptr_t align = (uintptr_t)pe | (uintptr_t)add_rows | (uintptr_t)head_weight | (uintptr_t)scratch | (uintptr_t)partials;
  for (int l = 0; l < 3; ++l) {
    OCOCC_REQUIRE(ln_weight[l] && ln_bias[l] && dz_out[l] && (l == 2 || w_t_frag[l]), "null pointer");
    OCOCC_REQUIRE(recompute ? (w_frag[l] && (l == 2 || y_out[l])) : (z_parked[l] && stats[l]), "null pointer");
    a.w[l] = recompute ? (const uint16_t*)w_frag[l] : nullptr;
    a.ln_w[l] = (const float*)ln_weight[l];
    a.ln_b[l] = (const float*)ln_bias[l];
    a.dz_out[l] = (uint16_t*)dz_out[l];
    a.z_in[l] = recompute ? nullptr : (const uint32_t*)z_parked[l];
    a.stats_in[l] = recompute ? nullptr : (const float*)stats[l];
    align |= (uintptr_t)ln_weight[l] | (uintptr_t)ln_bias[l] | (uintptr_t)dz_out[l] | (uintptr_t)a.w[l] | (uintptr_t)a.z_in[l];
    if (l < 2) {
      a.wt[l] = (const uint16_t*)w_t_frag[l];
      a.y_out[l] = recompute ? (uint16_t*)y_out[l] : nullptr;
      align |= (uintptr_t)w_t_frag[l] | (uintptr_t)a.y_out[l];
    }
    a.drop[l].thr = drop_threshold;
    a.drop[l].scale = drop_threshold ? 65536.f / (65536.f - (float)drop_threshold) : 1.f;
    a.drop[l].seed_lo = drop_threshold ? (uint32_t)dropout_seeds[l] : 0u;
    a.drop[l].seed_hi = drop_threshold ? (uint32_t)(dropout_seeds[l] >> 32) : 0u;
  }
  OCOCC_REQUIRE((align & 15) == 0, "pointers must be 16-byte aligned");
  a.pe = pe;
  a.add = add_rows;
  a.add_idx = add_index;
  a.head_w = head_weight;
  a.dlogit = dlogit;
  a.partials = partials;
  a.scratch = (uint32_t*)scratch;
  a.eps = eps;
  a.rows = rows;
  const unsigned grid = (unsigned)ococc_occ_mlp_bwd_workgroups(rows);
#define OCOCC_LAUNCH_BWD(DROP_, RC_)                                                                                      \
  do {                                                                                                                    \
    OCOCC_HIP(hipFuncSetAttribute((const void*)occ_mlp_bwd_kernel<DROP_, RC_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  kOccBwdLds));                                                                           \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(occ_mlp_bwd_kernel<DROP_, RC_>), dim3(grid), dim3(kThreads), kOccBwdLds,            \
                       (hipStream_t)stream, a);                                                                           \
  } while (0)
  if (drop_threshold) {
    if (recompute) OCOCC_LAUNCH_BWD(true, true); else OCOCC_LAUNCH_BWD(true, false);
  } else {
    if (recompute) OCOCC_LAUNCH_BWD(false, true); else OCOCC_LAUNCH_BWD(false, false);
  }
#undef OCOCC_LAUNCH_BWD
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
