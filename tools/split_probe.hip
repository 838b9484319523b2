// Can an fp32 GEMM K tile run faster as 3-way bf16 split products on the bf16 MFMA?  Register/LDS-only probe of ONE K tile's
// work per wave in the 64 x 320 NT tile (MI = 2 row tiles x NI = 10 column tiles per wave, 16 k):
//   fp32 : 80 x v_mfma_f32_16x16x4_f32
//   split: split the 12 fragments (4 floats per lane each) into hi / mid / lo bf16 (truncation split, exact), then
//          4 x v_mfma_f32_16x16x32_bf16 per (mi, ni):  [ah|am].[bh|bh] + [al|ah].[bh|bm] + [am|al].[bm|bm] + [ah|am].[bl|bl]
//          = 8 of the 9 cross terms (al.bl <= 2^-32 relative is dropped)
// Fragments come from LDS (ds_read_b128) so that the compiler cannot hoist the splits out of the loop.
// Build: hipcc --offload-arch=gfx950 -O3 split_probe.hip -o split_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hi16pack(float a, float b) {      // bf16(a) | bf16(b) << 16 by truncation
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
struct Split { unsigned h[2], m[2], l[2]; };      // 4 k values: hi, mid, lo as packed bf16 pairs
__device__ __forceinline__ Split split4(f32x4 v) {
  Split s;
  float r[4], q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float hi = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[i]) & 0xffff0000u);
    r[i] = v[i] - hi;
    const float mi = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r[i]) & 0xffff0000u);
    q[i] = r[i] - mi;
  }
  s.h[0] = hi16pack(v[0], v[1]); s.h[1] = hi16pack(v[2], v[3]);
  s.m[0] = hi16pack(r[0], r[1]); s.m[1] = hi16pack(r[2], r[3]);
  s.l[0] = hi16pack(q[0], q[1]); s.l[1] = hi16pack(q[2], q[3]);
  return s;
}
__device__ __forceinline__ bf16x8 cat(const unsigned* x, const unsigned* y) {
  const u32x4 u = {x[0], x[1], y[0], y[1]};
  return __builtin_bit_cast(bf16x8, u);
}

template <int MODE>
__global__ void __launch_bounds__(256, 3) probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[12 * 256 * 4 / 4 + 64 * 16];
  constexpr int MI = 2, NI = 10;
  for (int i = threadIdx.x; i < (int)(sizeof(lds) / 4); i += 256) lds[i] = 1.0f + 1e-3f * (i % 97);
  __syncthreads();
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    f32x4 a[MI], b[NI];
    const int base = ((it & 3) * 16 + lane) * 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(lds + base + mi * 256);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(lds + base + (2 + ni) * 256);
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ni][s], a[mi][s], acc[mi][ni], 0, 0, 0);
    } else {
      Split sa[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) sa[mi] = split4(a[mi]);
      bf16x8 a_hm[MI], a_lh[MI], a_ml[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) { a_hm[mi] = cat(sa[mi].h, sa[mi].m); a_lh[mi] = cat(sa[mi].l, sa[mi].h); a_ml[mi] = cat(sa[mi].m, sa[mi].l); }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const Split sb = split4(b[ni]);
        const bf16x8 b_hh = cat(sb.h, sb.h), b_hm = cat(sb.h, sb.m), b_mm = cat(sb.m, sb.m), b_ll = cat(sb.l, sb.l);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_hh, a_hm[mi], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_hm, a_lh[mi], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_mm, a_ml[mi], acc[mi][ni], 0, 0, 0);
          if (MODE == 1) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_ll, a_hm[mi], acc[mi][ni], 0, 0, 0);
        }
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) s += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char* name, int iters) {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 3;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  // fp32-equivalent flops: per wave and iteration a 32 x 160 x 16 product
  const double flops = (double)grid * 4 * iters * 2.0 * 32 * 160 * 16;
  printf("%-28s %.3f ms  %.1f fp32-equivalent TFLOP/s  (%.0f cycles per K tile and wave at 2.4 GHz, 3 waves per SIMD)\n", name, best,
         flops / best / 1e9, best * 1e-3 * 2.4e9 / iters / 3.0);
  hipFree(out);
}

int main() {
  run<0>("fp32 mfma 16x16x4", 4000);
  run<1>("bf16 split, 8 of 9 terms", 4000);
  run<2>("bf16 split, 6 of 9 terms", 4000);
  return 0;
}
