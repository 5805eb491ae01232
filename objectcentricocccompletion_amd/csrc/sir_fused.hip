// The one-launch SIRLayer, host side: which form a call takes, the barrier words per (device, stream), the tile size and
// LDS bytes of the launch.  The kernels are in csrc/sir_fused_impl.hpp (one translation unit per tile size).
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "point_mlp_tile.hpp"
#include "sir_fused.hpp"

namespace {

std::mutex g_bar_mutex;
std::map<std::pair<int, void*>, uint32_t*> g_bar_words;
int g_sir_fused = -1;   // -1: OCOCC_SIR_FUSED (default on); 0 / 1: pinned by ococc_sir_layer_set_fused
int g_grid_override = 0;               // tests: launch this many workgroups whatever the device holds (ococc_sir_layer_fused_debug)
uint64_t g_bar_ticks = 200000000ull;   // a barrier wait gives up after 2 s of the 100 MHz clock
uint32_t* g_err_host = nullptr;        // host-mapped: a wait of some one-launch layer gave up (written by the device, read here)

// The word the kernels write when a barrier wait gives up, in host memory the device can store to: the library looks at
// it (a plain host load, no synchronisation) in front of every one-launch layer.
int stranded(uint32_t gave_up) {
  g_sir_fused = 0;
  snprintf(ococc_err_buf, sizeof(ococc_err_buf),
           "ococc_sir_layer: a grid barrier of a one-launch SIR layer gave up (wait %u): not every workgroup of its persistent grid "
           "was resident -- another process or stream holds compute-unit slots of this device. That layer's results are incomplete; "
           "the per-block launches are used from now on (OCOCC_SIR_FUSED=0 selects them from the start)", gave_up - 1u);
  return OCOCC_ESTRANDED;
}

uint32_t* error_word() {
  std::lock_guard<std::mutex> lock(g_bar_mutex);
  if (g_err_host) return g_err_host;
  uint32_t* w = nullptr;
  if (hipHostMalloc((void**)&w, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return nullptr;
  *(volatile uint32_t*)w = 0u;
  g_err_host = w;
  return w;
}

// the barrier buffer of a stream (kernels of one stream run one after another; two streams must not share the words)
uint32_t* barrier_words(hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_bar_mutex);
  const auto key = std::make_pair(dev, (void*)stream);
  auto it = g_bar_words.find(key);
  if (it != g_bar_words.end()) return it->second;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;   // (no allocation under capture)
  uint32_t* w = nullptr;
  if (hipMalloc((void**)&w, kSirBarWords * sizeof(uint32_t)) != hipSuccess) return nullptr;
  if (hipMemset(w, 0, kSirBarWords * sizeof(uint32_t)) != hipSuccess) {
    (void)hipFree(w);
    return nullptr;
  }
  g_bar_words[key] = w;
  return w;
}

int fitting_signature(const SirFusedArgs& A) {
  for (int s = 0; s < kSirSignatures; ++s) {
    const SirSignature& S = kSirSignature[s];
    if (S.nr != A.nr || S.nv != A.nv) continue;
    bool fits = true;
    for (int b = 0; b < A.nr + A.nv; ++b) {
      const int kbw = (((A.b[b].k + 15) >> 4) + 3) / 4;
      fits = fits && nbw_of(A.b[b].n) == S.nbw[b] && kbw == S.kbw[b];
    }
    if (fits) return s;
  }
  return -1;
}

int launch(const SirFusedArgs& args, bool backward, hipStream_t stream) {
  const int sig = fitting_signature(args);
  if (sig < 0) return -1;
  SirFusedArgs A = args;
  A.bar = barrier_words(stream);
  if (!A.bar) return -1;
  uint32_t* err = error_word();
  if (!err) return -1;
  if (const uint32_t gave_up = *(volatile uint32_t*)err) {
    // An earlier one-launch layer could not gather its grid at barrier gave_up - 1: its maxima / gradients are incomplete.
    // Say so ONCE, loudly, and keep this process on the per-block launches from here on.
    *(volatile uint32_t*)err = 0u;
    return stranded(gave_up);
  }
  A.err_host = nullptr;
  if (hipHostGetDevicePointer((void**)&A.err_host, err, 0) != hipSuccess) return -1;
  A.bar_ticks = g_bar_ticks;
  A.census = nullptr;
  const int tile_rows = point_mlp_tile_rows(A.rows);
  int floats = 0;
  for (int b = 0; b < A.nr + A.nv; ++b) {
    const int f = lds_floats(pad_k(A.b[b].k), pad_k(A.b[b].n), tile_rows);
    floats = f > floats ? f : floats;
  }
  const int64_t tiles = ococc_cdiv(A.rows, tile_rows);
  // By default the one-launch form is taken while every tile has a workgroup of its own (on MI355X: <= 896 tiles of 32
  // rows): the blocks of a tile are latency-bound chains that want as many tiles in flight as the device holds, and a
  // persistent grid walking 1040 tiles with 1024 workgroups spends two rounds per phase (measured, whole configs[2] step:
  // 13.9 -> 13.2 ms at 4 tracklets = 256 tiles, but 21.7 -> 23.8 ms at 16 tracklets = 1040 tiles and 59.2 -> 61.6 ms at
  // 131 k rows = 4096 tiles).  ococc_sir_layer_set_fused(1) takes it at any size (tests).
  const bool one_tile_each = g_sir_fused != 1;
  if (tile_rows == 16) return sir_fused_launch_mb1(A, sig, backward, floats * 4, tiles, one_tile_each, g_grid_override, stream);
  if (tile_rows == 32) return sir_fused_launch_mb2(A, sig, backward, floats * 4, tiles, one_tile_each, g_grid_override, stream);
  return sir_fused_launch_mb4(A, sig, backward, floats * 4, tiles, one_tile_each, g_grid_override, stream);
}

}  // namespace

bool sir_fused_enabled() {
  if (g_sir_fused >= 0) return g_sir_fused != 0;
  static const int env = [] {
    const char* e = getenv("OCOCC_SIR_FUSED");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  return env != 0;
}

int sir_fused_forward(const SirFusedArgs& args, hipStream_t stream) { return launch(args, false, stream); }
int sir_fused_backward(const SirFusedArgs& args, hipStream_t stream) { return launch(args, true, stream); }

extern "C" int ococc_sir_layer_set_fused(int32_t mode) {
  OCOCC_REQUIRE(mode >= -1 && mode <= 1, "-1 (OCOCC_SIR_FUSED, default on), 0 (one launch per block) or 1 (one launch per layer)");
  g_sir_fused = mode;
  return OCOCC_OK;
}

extern "C" int ococc_sir_layer_fused_status(ococc_stream_t stream_, int32_t* status) {
  OCOCC_REQUIRE(status, "null pointer");
  *status = 0;
  uint32_t* w = barrier_words((hipStream_t)stream_);
  if (!w) return OCOCC_OK;
  uint32_t v = 0;
  OCOCC_HIP(hipStreamSynchronize((hipStream_t)stream_));
  OCOCC_HIP(hipMemcpy(&v, w + kSirBarError, sizeof(v), hipMemcpyDeviceToHost));
  *status = (int32_t)v;
  return OCOCC_OK;
}

// The check a caller makes where it is synchronised anyway (end of a step's read-back, checkpoint, end of a benchmark
// loop): no synchronisation, no copy -- a host load of the word the kernels write when a wait gives up.  Returns an error
// (and pins the per-block launches) when some one-launch layer of this process could not gather its grid.
extern "C" int ococc_sir_layer_fused_check(void) {
  uint32_t* err;
  {
    std::lock_guard<std::mutex> lock(g_bar_mutex);
    err = g_err_host;
  }
  if (!err) return OCOCC_OK;
  if (const uint32_t gave_up = *(volatile uint32_t*)err) {
    *(volatile uint32_t*)err = 0u;
    return stranded(gave_up);
  }
  return OCOCC_OK;
}

// Test hooks: grid > 0 launches that many workgroups whatever the device holds (a grid that cannot be resident at once
// strands at its first barrier); timeout_ms > 0 bounds a barrier wait (default 2000).  0 restores either default.
extern "C" int ococc_sir_layer_fused_debug(int32_t grid, int32_t timeout_ms) {
  OCOCC_REQUIRE(grid >= 0 && timeout_ms >= 0, "negative arguments");
  g_grid_override = grid;
  g_bar_ticks = timeout_ms > 0 ? (uint64_t)timeout_ms * 100000ull : 200000000ull;
  if (grid == 0 && timeout_ms == 0) {   // back to the defaults: forget what the forced time-outs left behind
    int dev = 0;
    OCOCC_HIP(hipGetDevice(&dev));
    OCOCC_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(g_bar_mutex);
    for (auto& kv : g_bar_words)
      if (kv.first.first == dev) OCOCC_HIP(hipMemset(kv.second + kSirBarError, 0, sizeof(uint32_t)));
    if (g_err_host) *(volatile uint32_t*)g_err_host = 0u;
  }
  return OCOCC_OK;
}
