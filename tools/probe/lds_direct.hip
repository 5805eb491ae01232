// Probe: where does global_load_lds_dwordx4 put each lane's 16 bytes?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const uint4* __restrict__ g, uint4* out) {
  __shared__ uint4 lds[512];
  const int wave = threadIdx.x >> 6;
  // every lane reads "its own" piece; wave w targets lds + 64*w
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + threadIdx.x),
                                   (void __attribute__((address_space(3)))*)(lds + 64 * wave), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = lds[threadIdx.x];
}
int main() {
  uint4 h[256], *d, *o;
  for (int i = 0; i < 256; ++i) h[i] = make_uint4(i, i + 1000, i + 2000, i + 3000);
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 256>>>(d, o);
  uint4 r[256];
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (r[i].x != (unsigned)i || r[i].w != (unsigned)i + 3000) { if (bad < 8) printf("pos %d holds %u %u %u %u\n", i, r[i].x, r[i].y, r[i].z, r[i].w); ++bad; }
  printf("bad=%d\n", bad);
  return 0;
}
