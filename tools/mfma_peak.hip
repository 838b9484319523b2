// Sustained fp32 MFMA rate of the device: every wave issues independent v_mfma_f32_16x16x4_f32 chains
// from registers only (no memory traffic), 4..12 waves per CU.  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
static void run(int wg_per_cu, int iters) {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1e-6f);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1e-6f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)grid * 4 * iters * NACC * 2048.0;
  printf("acc tiles/wave %2d, %d WG/CU (%d waves/SIMD): %.3f ms  %.1f TFLOP/s\n", NACC, wg_per_cu, wg_per_cu, best, flops / best / 1e9);
  hipFree(out);
}

int main() {
  run<4>(1, 20000); run<4>(2, 20000); run<8>(1, 10000); run<8>(2, 10000); run<8>(3, 10000); run<20>(3, 4000);
  // long run: does the rate hold once the part is hot?
  run<8>(2, 400000);
  return 0;
}
