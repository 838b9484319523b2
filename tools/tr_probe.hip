// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds bf16 words whose value = their element index.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(int mode, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr = 0;
  if (mode == 0) addr = 0;                         // uniform address
  else if (mode == 1) addr = 8u * l;               // every lane its own consecutive 8 bytes
  else if (mode == 2) addr = 32u * (l & 15) + 8u * (l >> 4);   // lane (i = l&15, g = l>>4): row i of a [16][16] bf16 matrix (32 B rows), 8-byte column group g
  else if (mode == 3) addr = 32u * (l >> 4) * 4 + 32u * ((l & 15) >> 2) + 8u * (l & 3);  // k-major [k][16 i]: group g rows 4g..4g+3
  addr += (unsigned)(uintptr_t)(void*)0;
  const unsigned base = (unsigned)(uintptr_t)lds;   // LDS byte address
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("  l%02d: %4d %4d %4d %4d", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l % 4 == 3) printf("\n"); }
  }
  return 0;
}
