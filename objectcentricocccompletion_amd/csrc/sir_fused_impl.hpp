// A whole SIRLayer per launch (csrc/sir_fused.hpp), for ONE tile size: csrc/sir_fused_mb{1,2,4}.hip define
// OCOCC_SIR_MB and include this file (three translation units, so that the tile bodies compile side by side).
//
// The grid is persistent (at most as many workgroups as the device holds at once) and walks the row tiles; per tile the
// blocks run back to back through the tile bodies of csrc/point_mlp_tile.hpp, a block's output rows going through the
// (L2-resident) slab the backward pass reads anyway.  Where a block needs the segment maxima of ALL tiles -- every vfe
// block after the first; in the backward pass the gradient those maxima collect, and the weight-gradient products --
// the grid meets at a barrier: one counter + one generation word per barrier in a per-stream buffer, the generation
// compared with an epoch word that the last barrier of a launch advances (nothing to reset between launches, nothing the
// host has to count, replayable from a graph).  A wait is bounded (2 s of the constant 100 MHz clock): a barrier that
// cannot complete leaves a mark in word kSirBarError instead of hanging the device.
//
// Which instantiation of a tile body a block runs on is a compile-time property (SirSignature): with the choice made at
// run time inside one kernel -- a switch over the 12 backward bodies -- the register allocator spilled 500-1100 VGPRs.
#pragma once
#include <mutex>
#include <type_traits>

#include "point_mlp_tile.hpp"
#include "sir_fused.hpp"

namespace {

struct GridBar {
  uint32_t* w;
  uint32_t target;
  uint32_t* err_host;   // host-mapped word: a wait that gave up (bar_wait)
  uint64_t ticks;       // how long a wait lasts at most (100 MHz clock)
};
__device__ __forceinline__ GridBar bar_begin(uint32_t* w, uint32_t* err_host, uint64_t ticks) {
  return GridBar{w, __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u, err_host, ticks};
}
// No cache maintenance at the barriers: a device-scope fence on this multi-die part writes back and invalidates the L2
// of the die (measured with one fence per workgroup and barrier: the forward launch 113 us where its five bodies take
// ~70 us as launches of their own; with every thread fencing after every body the 16-tracklet step went 22.4 -> 43.0
// ms).  Instead, everything one workgroup produces for ANOTHER of the same launch goes through device-scope atomics
// (the maxima, their arg-max rows, the gradient they collect; their start values are atomic stores) and is read back
// with device-scope atomic loads (load_shared_result<true>), both of which act past the non-coherent cache levels;
// everything else a phase reads was written by the same workgroup (workgroup scope: one compute unit, one cache) or by an
// earlier launch.  That is why the weight-gradient products -- which read every tile's dz and input rows -- stay a
// launch of their own.  A barrier is then: every wave drains its outstanding vector-memory operations -- an explicit
// s_waitcnt vmcnt(0): the workgroup barrier alone compiles to a bare s_barrier, which does NOT wait for the no-return
// atomics on the maxima / arg-max rows / collected gradients still in flight (round-5 advisor finding, checked on the
// built code object by tools/check_sir_barrier_isa.py); it is a counter wait, no cache write-back or invalidate -- then
// the workgroup's barrier, one arrival; the last arrival publishes the generation.
__device__ __forceinline__ void bar_arrive(const GridBar& g, int b, bool last_of_launch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t old = atomicAdd(&g.w[1 + 2 * b], 1u);
    if (old == gridDim.x - 1) {
      uint32_t seen = atomicExch(&g.w[1 + 2 * b], 0u);
      if (last_of_launch) seen += atomicExch(&g.w[0], g.target);
      if (seen != 0xffffffffu) atomicExch(&g.w[2 + 2 * b], g.target);   // (after the two above have returned)
    }
  }
}
// A wait that cannot complete (a workgroup of the grid is not resident: another process or stream holds its slot) gives
// up after `ticks` of the constant 100 MHz clock and says so in TWO places: the sticky device word kSirBarError
// (ococc_sir_layer_fused_status) and a host-mapped word that the library reads, without any synchronisation, in front
// of its next one-launch layer and turns into an error return (csrc/sir_fused.hip: the product path raises instead of
// training on incomplete maxima).  Once a launch is marked, its later waits give up at once.
__device__ __forceinline__ void bar_wait(const GridBar& g, int b) {
  const uint64_t ticks = g.ticks;
  uint32_t* err_host = g.err_host;
  if (threadIdx.x == 0) {
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load(&g.w[2 + 2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.target) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > ticks ||
          __hip_atomic_load(&g.w[kSirBarStranded], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g.target) {
        atomicExch(&g.w[kSirBarError], 1u + (uint32_t)b);
        atomicExch(&g.w[kSirBarStranded], g.target);
        if (err_host) __hip_atomic_store(err_host, 1u + (uint32_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
}
// How many workgroups of THIS kernel (its registers, its LDS) the device really holds at once: every workgroup announces
// itself, stays for 200 us, and the ones that leave before anybody else has left record how many had arrived -- all of
// those were resident together.  Launched once per (kernel, LDS size, device) with twice the grid the occupancy API
// promises (that API reads one workgroup per compute unit high for some register counts on this stack,
// MI355X_MICROARCH.md "Correctness boundaries": a grid sized by it alone strands a workgroup per CU at the first barrier).
// Foreign work on the device during the census can only lower the count.
__device__ __forceinline__ void census_body(uint32_t* c) {
  if (threadIdx.x == 0) {
    atomicAdd(&c[0], 1u);
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000ull) __builtin_amdgcn_s_sleep(8);
    const uint32_t arrived = __hip_atomic_load(&c[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t left = __hip_atomic_load(&c[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == 0) atomicMax(&c[2], arrived);
    atomicAdd(&c[1], 1u);
  }
}
// between two blocks of one tile: the rows a block wrote are read by other waves of the same workgroup (workgroup scope:
// the waves share the compute unit's cache), and the LDS tile changes hands
__device__ __forceinline__ void tile_sync() { __syncthreads(); }
__device__ __forceinline__ void store_shared_result(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_shared_result(int32_t* p, int32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// The layer's description is read through the kernel-argument segment pointer, and both that pointer and the thread
// index pass through an empty asm statement in front of every block body: the compiler must then recompute, per body and
// per tile, what it would otherwise compute once for all five bodies in front of the tile loop and keep alive -- the
// per-lane source addresses, the weight-fragment pointers, every pointer of the description (measured without: 490 VGPRs
// and 850 SGPRs spilled in the backward kernel, where the bodies alone use <= 137 VGPRs).
using KArgs = const __attribute__((address_space(4))) SirFusedArgs;
__device__ __forceinline__ KArgs* kernel_args() { return (KArgs*)__builtin_amdgcn_kernarg_segment_ptr(); }
__device__ __forceinline__ void launder(KArgs*& p, Tile& t) {
  asm volatile("" : "+s"(p));
  asm volatile("" : "+v"(t.tid));
}

template <int S>
struct Sig {
  static constexpr int nr = kSirSignature[S].nr, nv = kSirSignature[S].nv, nl = nr + nv;
  static constexpr int nbw(int q) { return kSirSignature[S].nbw[q]; }
  static constexpr int kbw(int q) { return kSirSignature[S].kbw[q]; }
};

template <class SIG, int Q>
__device__ __forceinline__ PointMlpIn sir_block_input(KArgs* A) {
  PointMlpIn in{};
  in.rows = A->rows;
  in.bscale = 1.f;
  if constexpr (Q < SIG::nr) {   // rel_mlp: f_cluster / rel_dist_scaler, then the block before
    in.a = Q == 0 ? A->fc : A->b[Q > 0 ? Q - 1 : 0].y;
    in.ka = A->b[Q].k;
    in.lda = Q == 0 ? A->cluster_cols : A->b[Q > 0 ? Q - 1 : 0].n;
    in.colscale = Q == 0 ? A->rel_cs : nullptr;
  } else if constexpr (Q == SIG::nr) {   // first vfe block: [features (*) gate (*) column scale | f_cluster * bscale]
    in.a = A->feats;
    in.ka = in.lda = A->feat_cols;
    in.colscale = A->col;
    if constexpr (SIG::nr > 0) {
      in.mul = A->b[SIG::nr > 0 ? SIG::nr - 1 : 0].y;
      in.ldm = A->b[SIG::nr > 0 ? SIG::nr - 1 : 0].n;
    } else if (A->gate) {
      in.mul = A->gate;
      in.ldm = A->ld_gate;
    }
    if (A->with_cc) {
      in.b = A->fc;
      in.kb = in.ldb = A->cluster_cols;
      in.bscale = A->bscale;
    }
    in.inv = A->inv;
  } else {   // later vfe blocks: [y of the block before | its maxima, gathered back]
    in.a = A->b[Q - 1].y;
    in.ka = in.lda = A->b[Q - 1].n;
    in.v = A->b[Q - 1].m;
    in.kv = A->b[Q - 1].n;
    in.inv = A->inv;
  }
  return in;
}

template <int MB, class SIG, int Q>
__device__ __forceinline__ void sir_forward_block(KArgs* A, Tile t) {
  launder(A, t);
  const PointMlpIn in = sir_block_input<SIG, Q>(A);
  auto& B = A->b[Q];
  point_mlp_fwd_tile<SIG::nbw(Q), MB, true>(in, B.wf, B.n, B.ln_w, B.ln_b, B.eps, B.act, B.y, Q >= SIG::nr ? B.m : nullptr, t);
  tile_sync();
}

// rows of the tile that attain their segment's maximum propose themselves; the smallest row stands
// (segment_argmax_kernel of csrc/point_mlp.hip, for the rows of one tile)
template <int MB>
__device__ __forceinline__ void sir_tile_argmax(const float* __restrict__ y, const float* __restrict__ m, int32_t* __restrict__ arg,
                                                int n, const int32_t* __restrict__ inv, int64_t rows, const Tile& t) {
  const int q4 = (n + 3) >> 2;
  for (int i = t.tid; i < 16 * MB * q4; i += kT) {
    const int64_t row = t.row0 + i / q4;
    if (row >= rows) break;
    const int c0 = (i % q4) * 4;
    const int seg = inv[row];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ch = c0 + c;
      if (ch < n && y[row * n + ch] == load_shared_result<true>(m + (int64_t)seg * n + ch)) atomicMin(arg + (int64_t)seg * n + ch, (int32_t)row);
    }
  }
}

template <int MB, class SIG>
__global__ void __launch_bounds__(kT, 2) sir_fused_fwd_kernel(SirFusedArgs) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  constexpr int TRM = 16 * MB;
  KArgs* A = kernel_args();
  const int64_t rows = A->rows, groups = A->groups;
  const int64_t tiles = (rows + TRM - 1) / TRM;
  if (A->census) {   // (once per kernel, LDS size and device: how many workgroups are resident together)
    census_body(A->census);
    return;
  }
  const GridBar gb = bar_begin(A->bar, A->err_host, A->bar_ticks);
  // maxima start at -inf, arg-max rows at "none"
  static_for<0, SIG::nv>([&](auto iv) {
    auto& B = A->b[SIG::nr + decltype(iv)::value];
    float* m = B.m;
    int32_t* arg = B.arg;
    const int64_t count = groups * B.n;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < count; e += (int64_t)gridDim.x * kT) {
      store_shared_result(m + e, -INFINITY);
      store_shared_result(arg + e, 0x7fffffff);
    }
  });
  bar_arrive(gb, 0, false);
  bool started = false;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const Tile t{tile, tile * TRM, smem_f, (int)threadIdx.x};
    static_for<0, SIG::nr>([&](auto j) { sir_forward_block<MB, SIG, decltype(j)::value>(A, t); });   // gate = rel_mlp(f_cluster / rel_dist_scaler)
    if (!started) {   // (every workgroup initialised its share long ago: the rel blocks of a tile lie in between)
      bar_wait(gb, 0);
      started = true;
    }
    sir_forward_block<MB, SIG, SIG::nr>(A, t);
  }
  if (!started) bar_wait(gb, 0);
  bar_arrive(gb, 1, SIG::nv == 1);
  static_for<1, SIG::nv>([&](auto iv) {
    constexpr int i = decltype(iv)::value;
    bar_wait(gb, i);   // the maxima of block i - 1 are final
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
      const Tile t{tile, tile * TRM, smem_f, (int)threadIdx.x};
      auto& P = A->b[SIG::nr + i - 1];
      sir_tile_argmax<MB>(P.y, P.m, P.arg, P.n, A->inv, rows, t);
      sir_forward_block<MB, SIG, SIG::nr + i>(A, t);
    }
    bar_arrive(gb, i + 1, i + 1 == SIG::nv);
  });
  bar_wait(gb, SIG::nv);   // (the launch's last barrier: the epoch word has moved on) the last block's maxima are final
  auto& Z = A->b[SIG::nl - 1];
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const Tile t{tile, tile * TRM, smem_f, (int)threadIdx.x};
    sir_tile_argmax<MB>(Z.y, Z.m, Z.arg, Z.n, A->inv, rows, t);
    if (A->shortcut) {   // out = y + the non-xyz feature columns
      const int n = Z.n, ldf = A->feat_cols;
      const float* y = Z.y;
      const float* feats = A->feats;
      float* out = A->y_out;
      for (int i = threadIdx.x; i < TRM * n; i += kT) {
        const int64_t row = t.row0 + i / n;
        if (row >= rows) break;
        const int c = i % n;
        out[row * n + c] = y[row * n + c] + feats[row * ldf + 3 + c];
      }
    }
  }
  {   // the concatenated maxima
    const int sum_n = A->sum_n;
    const int64_t count = groups * sum_n;
    float* out = A->groups_out;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < count; e += (int64_t)gridDim.x * kT) {
      const int64_t g = e / sum_n;
      int c = (int)(e - g * sum_n), i = 0;
      while (i + 1 < SIG::nv && c >= A->b[SIG::nr + i].n) c -= A->b[SIG::nr + i++].n;
      out[e] = load_shared_result<true>(A->b[SIG::nr + i].m + g * A->b[SIG::nr + i].n + c);
    }
  }
}

template <int MB, class SIG, int Q>
__device__ __forceinline__ void sir_backward_block(KArgs* A, const float* dy, int ldy, const float* dvmax, int ldvm, const float* dvmax2,
                                                   float* da, float* dmul, float* dv, Tile t) {
  launder(A, t);
  const PointMlpIn in = sir_block_input<SIG, Q>(A);
  auto& B = A->b[Q];
  const int32_t* arg = (dvmax || dvmax2) ? B.arg : nullptr;
  point_mlp_bwd_tile<SIG::nbw(Q), SIG::kbw(Q), MB, true>(in, B.wf, B.wtf, B.n, B.ln_w, B.ln_b, B.eps, B.act, dy, ldy, dvmax, ldvm, dvmax2, arg,
                                                   B.dz, B.xcat, da, dmul, nullptr, dv, B.lnp, t);
  tile_sync();
}

template <int MB, class SIG>
__global__ void __launch_bounds__(kT, 2) sir_fused_bwd_kernel(SirFusedArgs) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  constexpr int TRM = 16 * MB;
  constexpr int nr = SIG::nr, nv = SIG::nv, nl = SIG::nl;
  KArgs* A = kernel_args();
  const int64_t rows = A->rows, groups = A->groups;
  const int64_t tiles = (rows + TRM - 1) / TRM;
  if (A->census) {
    census_body(A->census);
    return;
  }
  const GridBar gb = bar_begin(A->bar, A->err_host, A->bar_ticks);
  // the gradients the gathered maxima collect (float atomics at run ends) start at zero
  static_for<1, nv>([&](auto iv) {
    constexpr int i = decltype(iv)::value;
    const int64_t count = groups * A->b[nr + i - 1].n;
    float* dv = A->b[nr + i].dv;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < count; e += (int64_t)gridDim.x * kT) store_shared_result(dv + e, 0.f);
  });
  bar_arrive(gb, 0, false);
  // vfe blocks from the last one down: block i's gradient of its gathered maxima is complete only when every tile has
  // added its part -- block i - 1 starts behind a barrier
  static_for<0, nv - 1>([&](auto ii) {
    constexpr int i = nv - 1 - decltype(ii)::value, q = nr + i;
    bar_wait(gb, nv - 1 - i);
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
      const Tile t{tile, tile * TRM, smem_f, (int)threadIdx.x};
      int off_m = 0;
#pragma unroll
      for (int b = nr; b < q; ++b) off_m += A->b[b].n;
      const float* d_groups = A->d_groups;
      const float* dy_cur = i == nv - 1 ? A->dy : A->b[i == nv - 1 ? q : q + 1].da;
      const float* carry = i == nv - 1 ? nullptr : A->b[i == nv - 1 ? q : q + 1].dv;
      sir_backward_block<MB, SIG, q>(A, dy_cur, i == nv - 1 ? A->ld_dy : A->b[q].n, d_groups ? d_groups + off_m : nullptr, A->ld_dg, carry, A->b[q].da, nullptr,
                                     A->b[q].dv, t);
    }
    bar_arrive(gb, nv - i, false);
  });
  bar_wait(gb, nv - 1);
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const Tile t{tile, tile * TRM, smem_f, (int)threadIdx.x};
    {
      const float* dy_cur = nv == 1 ? A->dy : A->b[nv == 1 ? nr : nr + 1].da;
      const float* carry = nv == 1 ? nullptr : A->b[nv == 1 ? nr : nr + 1].dv;
      sir_backward_block<MB, SIG, nr>(A, dy_cur, nv == 1 ? A->ld_dy : A->b[nr].n, A->d_groups, A->ld_dg, carry, A->dfeat,
                                      (nr || A->gate) ? A->dgate : nullptr, nullptr, t);
    }
    if (A->shortcut && A->dfeat && A->dy) {   // dfeat[:, 3:] += dy
      const int n = A->b[nl - 1].n, ldf = A->feat_cols, ldy = A->ld_dy;
      float* dfeat = A->dfeat;
      const float* dy = A->dy;
      for (int i = threadIdx.x; i < TRM * n; i += kT) {
        const int64_t row = t.row0 + i / n;
        if (row >= rows) break;
        const int c = i % n;
        dfeat[row * ldf + 3 + c] += dy[row * ldy + c];
      }
    }
    // (every block writes the gradient of its input rows to a buffer of its own: the blocks' row widths differ, so in a
    // shared buffer one tile's rows of one block would lie on another tile's rows of another block)
    static_for<0, nr>([&](auto jj) {
      constexpr int j = nr - 1 - decltype(jj)::value;
      const float* dgate = j == nr - 1 ? A->dgate : A->b[j == nr - 1 ? j : j + 1].da;
      sir_backward_block<MB, SIG, j>(A, dgate, A->b[j].n, nullptr, 0, nullptr, j > 0 ? A->b[j].da : nullptr, nullptr, nullptr, t);
    });
  }
  bar_arrive(gb, nv, true);   // (nobody waits: the launch's last arrival advances the epoch word)
}

constexpr int kImplMaxDevices = 64;
std::mutex g_launch_mutex;   // the per-kernel caches below (ctypes callers may hold several host threads)

// The persistent grid of a kernel: min(what the occupancy API promises, what a census launch of the SAME kernel with the
// SAME LDS size saw resident together) less an eighth (a collective's kernels launched from gradient hooks run beside the
// backward pass and must find room while this grid spins at a barrier).  The form assumes ONE one-launch layer in flight
// per device: two persistent grids (two processes, two streams) can hold each other's slots -- then a bounded wait gives
// up and the library reports it (csrc/sir_fused.hip).
template <typename K>
int persistent_launch(K kernel, const SirFusedArgs& A, int lds, int64_t tiles, bool one_tile_each, int grid_override,
                      hipStream_t stream, int* lds_set, int* cached_lds, int* cached_cap) {
  int device = 0;
  OCOCC_HIP(hipGetDevice(&device));
  const int dev = device % kImplMaxDevices;
  std::lock_guard<std::mutex> lock(g_launch_mutex);
  if (lds > lds_set[dev]) {
    OCOCC_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    lds_set[dev] = lds;
  }
  if (cached_lds[dev] != lds) {   // how many workgroups the device holds at once: every one of them must be running
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return -1;   // (the census reads back)
    int per_cu = 0, cus = 0;
    OCOCC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kT, (size_t)lds));
    OCOCC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    OCOCC_REQUIRE(per_cu >= 1 && cus >= 1, "the kernel does not fit a compute unit");
    const int promised = per_cu * cus;
    SirFusedArgs C = A;
    C.census = A.bar + kSirBarCensus;
    OCOCC_HIP(hipMemsetAsync(C.census, 0, 3 * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(kernel, dim3(2u * (unsigned)promised), dim3(kT), lds, stream, C);
    OCOCC_CHECK_LAUNCH();
    uint32_t seen = 0;
    OCOCC_HIP(hipMemcpyAsync(&seen, C.census + 2, sizeof(seen), hipMemcpyDeviceToHost, stream));
    OCOCC_HIP(hipStreamSynchronize(stream));
    OCOCC_REQUIRE(seen >= 1, "the residency census of the one-launch SIR layer saw no workgroup");
    const int resident = (int)seen < promised ? (int)seen : promised;
    cached_lds[dev] = lds;
    cached_cap[dev] = resident - resident / 8 > 0 ? resident - resident / 8 : 1;
  }
  if (one_tile_each && tiles > cached_cap[dev]) return -1;   // (see sir_fused.hip: more tiles than resident workgroups)
  unsigned grid = (unsigned)(tiles < cached_cap[dev] ? tiles : cached_cap[dev]);
  // (tests: a grid the device cannot hold at once -- the barrier must strand and the library must say so)
  if (grid_override > 0) grid = (unsigned)grid_override;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kT), lds, stream, A);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

}  // namespace

#define OCOCC_SIR_CAT_(a, b) a##b
#define OCOCC_SIR_CAT(a, b) OCOCC_SIR_CAT_(a, b)

int OCOCC_SIR_CAT(sir_fused_launch_mb, OCOCC_SIR_MB)(const SirFusedArgs& args, int signature, bool backward, int lds,
                                                    int64_t tiles, bool one_tile_each, int grid_override, hipStream_t stream) {
  OCOCC_REQUIRE(signature >= 0 && signature < kSirSignatures, "unknown block signature");
  static int lds_set[2 * kSirSignatures][kImplMaxDevices] = {}, c_lds[2 * kSirSignatures][kImplMaxDevices] = {},
             c_cap[2 * kSirSignatures][kImplMaxDevices] = {};
  const int slot = 2 * signature + (backward ? 1 : 0);
#define OCOCC_SIR_GO(S)                                                                                              \
  (backward ? persistent_launch(sir_fused_bwd_kernel<OCOCC_SIR_MB, Sig<S>>, args, lds, tiles, one_tile_each, grid_override, stream, \
                                lds_set[slot], c_lds[slot], c_cap[slot])                                             \
            : persistent_launch(sir_fused_fwd_kernel<OCOCC_SIR_MB, Sig<S>>, args, lds, tiles, one_tile_each, grid_override, stream, \
                                lds_set[slot], c_lds[slot], c_cap[slot]))
  static_assert(kSirSignatures == 4, "one case per signature");
  switch (signature) {
    case 0: return OCOCC_SIR_GO(0);
    case 1: return OCOCC_SIR_GO(1);
    case 2: return OCOCC_SIR_GO(2);
    default: return OCOCC_SIR_GO(3);
  }
#undef OCOCC_SIR_GO
}
