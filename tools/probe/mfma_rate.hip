// hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Issue rate of the matrix instructions the per-point SIR kernel could use, 8 independent accumulators per wave,
// 1 / 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int KIND>
__global__ void __launch_bounds__(256) rate(float* out, int iters) {
  f32x4 acc[8];
  f32x16 big[2], big4[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big4[i][j] = 0;
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0;
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  bf16x8 ab, bb;
  for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)a; bb[j] = (__bf16)b; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
      if (KIND == 2) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[i & 1], 0, 0, 0);
      if (KIND == 3) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, big[i & 1], 0, 0, 0);
      if (KIND == 4) big4[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, big4[i & 3], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  s += big[0][0] + big[1][5] + big4[0][1] + big4[1][2] + big4[2][3] + big4[3][4];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4 << 20);
  const int iters = 4000;
  const char* names[5] = {"f32 16x16x4 ", "bf16 16x16x32", "f32 32x32x2 ", "bf16 32x32x16 (2 acc)", "bf16 32x32x16 (4 acc)"};
  const double flops[5] = {2048, 16384, 4096, 32768, 32768};
  for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu)
    for (int kind = 0; kind < 5; ++kind) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters);
        if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters);
        if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters);
        if (kind == 3) hipLaunchKernelGGL(rate<3>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters);
        if (kind == 4) hipLaunchKernelGGL(rate<4>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double n = 256.0 * wg_per_cu * 4 * iters * 8;   // wave-level MFMAs
      printf("%s  %d waves/SIMD: %.3f ms, %.1f TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", names[kind], wg_per_cu, ms,
             n * flops[kind] / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (n / 1024));
    }
  return 0;
}
