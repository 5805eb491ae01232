// hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Issue rate of the matrix instructions the kernels of this repository use: 8 independent accumulators per wave, 1 / 2
// waves per SIMD, random-ish operands, cycles from the SHADER CLOCK (s_memtime) -- not from wall time at an assumed
// frequency: the chip lowers its clock under a dense matrix load.
//
// Round 3's version of this probe used the builtins on an accumulator ARRAY inside a runtime loop; hipcc rotated the
// accumulators through v_accvgpr_read / _write / _mov between iterations (~40 moves per 8 matrix instructions) and the
// probe read "35 cycles per 16x16x32 = half rate", which DESIGN 3.7 of round 3 built on.  The counters of the real kernels
// (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA = 16.0 for 16x16x32, 32.0 for 32x32x16) and MI355X_MICROARCH.md say 16 / 32:
// equal FLOP per clock.  Here every instruction is one asm statement on a register the compiler cannot move.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define MFMA16(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ab), "v"(bb))
#define MFMA32(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ab), "v"(bb))
#define MFMAF(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int KIND>
__global__ void __launch_bounds__(256) rate(float* out, unsigned long long* cyc, int iters) {
  float a = (threadIdx.x % 61) * 1e-3f - 0.03f, b = (threadIdx.x % 53) * 2e-3f - 0.05f;
  bf16x8 ab, bb;
  for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(a * (j + 1)); bb[j] = (__bf16)(b * (8 - j)); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  f32x16 d0, d1, d2, d3;
  for (int j = 0; j < 16; ++j) { d0[j] = 0; d1[j] = 0; d2[j] = 0; d3[j] = 0; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) { MFMAF(c0); MFMAF(c1); MFMAF(c2); MFMAF(c3); MFMAF(c4); MFMAF(c5); MFMAF(c6); MFMAF(c7); }
    if (KIND == 1) { MFMA16(c0); MFMA16(c1); MFMA16(c2); MFMA16(c3); MFMA16(c4); MFMA16(c5); MFMA16(c6); MFMA16(c7); }
    if (KIND == 2) { MFMA32(d0); MFMA32(d1); MFMA32(d2); MFMA32(d3); MFMA32(d0); MFMA32(d1); MFMA32(d2); MFMA32(d3); }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3] + d0[0] + d1[5] + d2[9] + d3[13];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  float* out;
  unsigned long long *cyc, *h = new unsigned long long[4096];
  hipMalloc(&out, 4 << 20);
  hipMalloc(&cyc, 4096 * 8);
  const int iters = 4000;
  const char* names[3] = {"f32 16x16x4  ", "bf16 16x16x32", "bf16 32x32x16"};
  const double flops[3] = {2048, 16384, 32768};
  for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu)
    for (int kind = 0; kind < 3; ++kind) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, cyc, iters);
        if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, cyc, iters);
        if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, cyc, 256 * wg_per_cu * 4 * 8, hipMemcpyDeviceToHost);
      double mean = 0;
      for (int i = 0; i < 256 * wg_per_cu * 4; ++i) mean += (double)h[i];
      mean /= 256.0 * wg_per_cu * 4;
      const double n = 256.0 * wg_per_cu * 4 * iters * 8;   // wave-level matrix instructions
      // (a wave's own cycles for its iters * 8 instructions; waves sharing a SIMD take turns, so the SIMD's interval is
      // that divided by the waves per SIMD; cycles / wall time = the clock the chip held)
      printf("%s  %d wave(s)/SIMD: %.1f shader cycles per instruction and SIMD, %.3f ms, %.0f TFLOP/s chip-wide, clock held %.2f GHz\n",
             names[kind], wg_per_cu, mean / (iters * 8.0) / wg_per_cu, ms, n * flops[kind] / (ms * 1e-3) / 1e12,
             mean / (ms * 1e-3) / 1e9);
    }
  return 0;
}
