// Does a saturating fp32-MFMA stream keep its rate while HBM is streamed at full speed beside it?
// Two kernels on two streams: a register-only MFMA loop (2 waves/SIMD) and a device copy (1 GiB).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_hbm_overlap.hip -o mfma_hbm_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters) {
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + threadIdx.x * 1e-9f, b = 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) copy_kernel(const float4* __restrict__ x, float4* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = x[i];
}

int main() {
  const size_t n4 = (size_t)64 << 20;            // 1 GiB
  float4 *x, *y; float* out;
  hipMalloc(&x, n4 * 16); hipMalloc(&y, n4 * 16); hipMalloc(&out, 4);
  hipMemset(x, 0, n4 * 16);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t a0, a1, b0, b1;
  hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
  const int iters = 60000, grid_m = 256 * 2, reps_c = 12, grid_c = 256 * 4;
  const double mflops = (double)grid_m * 4 * iters * 16 * 2048.0;
  auto run = [&](bool with_mfma, bool with_copy) {
    hipDeviceSynchronize();
    // the copy stream is enqueued first and runs many short launches: the MFMA kernel joins a busy device
    if (with_copy) { hipEventRecord(b0, s2); for (int r = 0; r < reps_c * (with_mfma ? 6 : 1); ++r) hipLaunchKernelGGL(copy_kernel, dim3(grid_c), dim3(256), 0, s2, x, y, n4); hipEventRecord(b1, s2); }
    if (with_mfma) { hipEventRecord(a0, s1); hipLaunchKernelGGL(mfma_loop, dim3(grid_m), dim3(256), 0, s1, out, iters); hipEventRecord(a1, s1); }
    hipDeviceSynchronize();
    float ma = 0.f, mb = 0.f;
    if (with_mfma) hipEventElapsedTime(&ma, a0, a1);
    if (with_copy) hipEventElapsedTime(&mb, b0, b1);
    printf("%-14s", with_mfma && with_copy ? "both together" : (with_mfma ? "MFMA alone" : "copy alone"));
    if (with_mfma) printf("  MFMA %8.2f ms %7.1f TFLOP/s", ma, mflops / ma / 1e9);
    if (with_copy) printf("  copy %8.2f ms %7.2f TB/s (read+write)", mb, 2.0 * reps_c * (with_mfma ? 6 : 1) * n4 * 16 / mb / 1e9);
    printf("\n");
  };
  run(true, false); run(true, false); run(false, true); run(false, true); run(true, true); run(true, true);
  return 0;
}
