// Standalone prototype of the LDS-DMA ("glds") fp32 MFMA NT GEMM main loop:  C[M][N] = A[M][K] . B[N][K]^T
// Both operands contraction-contiguous; LDS image [row][16 k] (64 B rows, XOR-swizzled 16-B chunks) filled by
// buffer_load_dwordx4 ... lds; fragments by ds_read_b128 (one read = the 4 k-steps of a K tile for one 16-row tile);
// fragments software-pipelined across the per-tile barrier.  Build: hipcc --offload-arch=gfx950 -O3 -o glds_proto glds_proto.hip
// Variants by -D: STAGES (2|3), NOPIPE (no cross-barrier fragment pipeline), WAVES_EU (launch bound).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include <algorithm>

#ifndef EPI
#define EPI 0
#endif
#ifndef STAGES
#define STAGES 2
#endif
#ifndef WAVES_EU
#define WAVES_EU 3
#endif

#ifdef NOSB
#define SBAR()
#else
#define SBAR() __builtin_amdgcn_sched_barrier(0)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int gsw(int j) { return (0x78 >> (2 * j)) & 3; }     // {0,2,3,1}

template <int WM, int WN, int NI, int MI>
__global__ void __launch_bounds__(WM * WN * 64, WAVES_EU)
glds_gemm(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
          int M, int N, int K, const float* __restrict__ in0, const float* __restrict__ in1, float* __restrict__ out1) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = WM * WN;
  constexpr int BM = 16 * MI * WM, BN = 16 * NI * WN, BK = 16;
  constexpr int NAI = BM / 16, NBI = BN / 16;
  constexpr int SA = (NAI + NW - 1) / NW, SB = (NBI + NW - 1) / NW;
  constexpr int STAGE = (BM + BN) * 64;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int NH = NI / 2;                    // B fragment batch size (X: ni < NH, Y: the rest)
#ifndef LDSPAD
#define LDSPAD 0
#endif
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * STAGE + LDSPAD];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int wrow = wm * 16 * MI, wcol = wn * 16 * NI;
  const int l15 = lane & 15, q = lane >> 4;

  // DMA slots of this wave
  const int drow = lane >> 2;
  const int dkq = (lane & 3) ^ gsw((lane >> 4) & 3);
  unsigned a_vo[SA], b_vo[SB];
  const int ntiles = (M + BM - 1) / BM;
#ifdef STAGGER
  {
    const int j = blockIdx.x >> 3;
    const int cls = (STAGGER == 1) ? (j / 32) % 3 : j % 3;
    if ((int)blockIdx.x < 768) for (int d = 0; d < cls * STAG_UNITS; ++d) __builtin_amdgcn_s_sleep(127);
  }
#endif
  auto calc_avo = [&](int tile_) __attribute__((always_inline)) {
    const int m0_ = tile_ * BM;
#pragma unroll
    for (int j = 0; j < SA; ++j) {
      const int ia = wave + NW * j;
      const int gm = m0_ + 16 * ia + drow;
      a_vo[j] = ((NW * (j + 1) <= NAI || ia < NAI) && gm < M) ? ((unsigned)gm * (unsigned)lda * 4u + (unsigned)dkq * 16u) : OOB;
    }
  };
#pragma unroll
  for (int j = 0; j < SB; ++j) {
    const int ib = wave + NW * j;
    const int n = 16 * ib + drow;
    b_vo[j] = ((NW * (j + 1) <= NBI || ib < NBI) && n < N) ? ((unsigned)n * (unsigned)ldb * 4u + (unsigned)dkq * 16u) : OOB;
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int m0 = tile * BM;
#ifdef MIXK
  const int Kt = ((tile / MIXK) & 1) ? K / 2 : K + K / 2;
#else
  const int Kt = K;
#endif
#ifdef MIXWG
  const bool only_epi = (((tile >> 3) / 32) & 1) != 0;      // alternates every 256 tiles: every CU sees both kinds
  const int T = only_epi ? 1 : (Kt + BK - 1) / BK;
#else
  const int T = (Kt + BK - 1) / BK;
#endif
#ifdef XPF
  if (tile == (int)blockIdx.x) calc_avo(tile);      // later tiles: offsets + first K tile were issued during the previous epilogue
#else
  if (tile != (int)blockIdx.x) __syncthreads();     // previous tile's epilogue no longer reads the stage memory
  calc_avo(tile);
#endif
  const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
  const rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);

  auto dma_tile = [&](int t, int st) __attribute__((always_inline)) {
    const int k0 = t * BK;
    const bool kok = 4 * dkq < Kt - k0;
    unsigned char* sb = smem + st * STAGE;
#pragma unroll
    for (int j = 0; j < SA; ++j) {
      const int ia = wave + NW * j;
      if (NW * (j + 1) <= NAI || ia < NAI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(sb + ia * 1024), 16,
                                                 kok ? a_vo[j] : OOB, k0 * 4, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < SB; ++j) {
      const int ib = wave + NW * j;
      if (NW * (j + 1) <= NBI || ib < NBI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(sb + BM * 64 + ib * 1024), 16,
                                                 kok ? b_vo[j] : OOB, k0 * 4, 0, 0);
    }
  };

  const int fsl = q ^ gsw((l15 >> 2) & 3);
  const unsigned a_fo = (unsigned)((wrow + l15) * 4 + fsl) * 16u;
  const unsigned b_fo = (unsigned)(BM * 64) + (unsigned)((wcol + l15) * 4 + fsl) * 16u;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 aC[MI], aP[MI], bX[NH], bY[NI - NH];

  auto read_a = [&](int st, f32x4* a) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(smem + st * STAGE + a_fo + mi * 1024);
  };
  auto read_b = [&](int st, f32x4* b, int ni0, int cnt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (i < cnt) b[i] = *reinterpret_cast<const f32x4*>(smem + st * STAGE + b_fo + (ni0 + i) * 1024);
  };
  auto mma = [&](const f32x4* a, const f32x4* b, int ni0, int cnt) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < NI; ++i)
        if (i < cnt)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[mi][ni0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i][s], a[mi][s], acc[mi][ni0 + i], 0, 0, 0);
  };

#if STAGES == 2
#ifdef XPF
  if (tile == (int)blockIdx.x) dma_tile(0, 0);
#else
  dma_tile(0, 0);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef NOPIPE
  for (int t = 0; t < T; ++t) {
    const int st = t & 1;
    if (t + 1 < T) dma_tile(t + 1, st ^ 1);
    read_a(st, aC);
    read_b(st, bX, 0, NH);
    read_b(st, bY, NH, NI - NH);
    mma(aC, bX, 0, NH);
    mma(aC, bY, NH, NI - NH);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#else
  for (int t = 0; t < T; ++t) {
    const int st = t & 1;
    if (t + 1 < T) dma_tile(t + 1, st ^ 1);
    read_a(st, aC);
    read_b(st, bX, 0, NH);
    SBAR();
    if (t > 0) mma(aP, bY, NH, NI - NH);          // leftovers of tile t-1 cover the LDS latency of the reads above
    SBAR();
    read_b(st, bY, NH, NI - NH);
    SBAR();
    mma(aC, bX, 0, NH);
    SBAR();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) aP[mi] = aC[mi];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  mma(aP, bY, NH, NI - NH);
#endif
#else   // 3 stages: DMA two tiles ahead, counted vmcnt, raw barrier
  constexpr int PER = SA + SB;   // DMA instructions per tile per wave (upper bound; all waves issue the same count when NAI,NBI % NW == 0)
  dma_tile(0, 0);
  if (T > 1) dma_tile(1, 1);
  if (T > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int st = 0;
  for (int t = 0; t < T; ++t) {
    const int st2 = st >= 1 ? st - 1 : 2;       // (st + 2) % 3
    if (t + 2 < T) dma_tile(t + 2, st2);
    read_a(st, aC);
    read_b(st, bX, 0, NH);
    SBAR();
    if (t > 0) mma(aP, bY, NH, NI - NH);
    SBAR();
    read_b(st, bY, NH, NI - NH);
    SBAR();
    mma(aC, bX, 0, NH);
    SBAR();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) aP[mi] = aC[mi];
    if (t + 2 < T) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    st = st == 2 ? 0 : st + 1;
  }
  mma(aP, bY, NH, NI - NH);
#endif

#ifdef MIXWG
  if (!only_epi) { float sx = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) sx += acc[mi][ni][0];
    if (sx == 12345.678f) C[0] = sx;
    continue; }
#endif
  // epilogue variants: EPI 0 = plain store, 1 = h-gate like (read in0, in1; write C, out1); 2 = none (one value per WG)
  // ELDS: stage the tile through LDS and stream whole rows (linear, 1 KiB per wave instruction) instead of fragment-shaped
  // 16 rows x 64 B accesses
#ifndef EPI
#define EPI 0
#endif
#if EPI == 2
  float s = 0.f;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) s += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
  if (s == 12345.678f) C[0] = s;
#elif !defined(ELDS)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int row = m0 + wrow + mi * 16 + l15;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = wcol + ni * 16 + 4 * q;
      if (row < M && col < N) {
        const size_t o = (size_t)row * ldc + col;
        float4 v = make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
#if EPI == 1
        const float4 z = *reinterpret_cast<const float4*>(in0 + o);
        const float4 x = *reinterpret_cast<const float4*>(in1 + o);
        const float4 h = make_float4(tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w));
        *reinterpret_cast<float4*>(C + o) = h;
        *reinterpret_cast<float4*>(out1 + o) = make_float4(h.x * z.x + x.x * (1.f - z.x), h.y * z.y + x.y * (1.f - z.y), h.z * z.z + x.z * (1.f - z.z), h.w * z.w + x.w * (1.f - z.w));
#else
        *reinterpret_cast<float4*>(C + o) = v;
#endif
      }
    }
  }
#else
  {
    constexpr int PITCH = BN + 4;                 // floats; rows 16 B aligned, 8 consecutive rows cover all 32 banks
#ifdef XPF
    // cross-tile prefetch: both stages are free after the K loop; the NEXT tile's first K tile goes to stage 0 now and
    // lands underneath this epilogue, which stages its rows in stage 1
    static_assert(16 * WM * PITCH * 4 <= STAGE, "epilogue pass fits ONE stage");
    float* ep = reinterpret_cast<float*>(smem + STAGE);
    if (tile + (int)gridDim.x < ntiles) { calc_avo(tile + gridDim.x); dma_tile(0, 0); }
#else
    static_assert(16 * WM * PITCH * 4 <= STAGES * STAGE, "epilogue tile fits the stage memory");
    float* ep = reinterpret_cast<float*>(smem);
#endif
    const int N4 = N / 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      __syncthreads();                            // stage memory (or the previous pass) no longer read
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        *reinterpret_cast<f32x4*>(ep + (wm * 16 + l15) * PITCH + wcol + ni * 16 + 4 * q) = acc[mi][ni];
      __syncthreads();
      // the pass holds rows {wm*16*MI + mi*16 + r : r < 16} for each wm: WM blocks of 16 consecutive global rows
#pragma unroll
      for (int b = 0; b < WM; ++b) {
        const int grow0 = m0 + b * 16 * MI + mi * 16;
        const int nrows = min(16, M - grow0);
        for (int i = tid; i < nrows * N4; i += NW * 64) {
          const int r = i / N4, c = i - r * N4;
          const f32x4 v = *reinterpret_cast<const f32x4*>(ep + (b * 16 + r) * PITCH + 4 * c);
          const size_t o = (size_t)(grow0 + r) * ldc + 4 * c;
#if EPI == 1
          const float4 z = *reinterpret_cast<const float4*>(in0 + o);
          const float4 x = *reinterpret_cast<const float4*>(in1 + o);
          const float4 h = make_float4(tanhf(v[0]), tanhf(v[1]), tanhf(v[2]), tanhf(v[3]));
          *reinterpret_cast<float4*>(C + o) = h;
          *reinterpret_cast<float4*>(out1 + o) = make_float4(h.x * z.x + x.x * (1.f - z.x), h.y * z.y + x.y * (1.f - z.y), h.z * z.z + x.z * (1.f - z.z), h.w * z.w + x.w * (1.f - z.w));
#elif EPI == 3
          const float4 h = make_float4(tanhf(v[0]), tanhf(v[1]), tanhf(v[2]), tanhf(v[3]));
          *reinterpret_cast<f32x4*>(ep + (b * 16 + r) * PITCH + 4 * c) = f32x4{h.x * 0.5f + v[1], h.y * 0.5f + v[2], h.z * 0.5f + v[3], h.w * 0.5f + v[0]};
          (void)o;
#else
          *reinterpret_cast<float4*>(C + o) = make_float4(v[0], v[1], v[2], v[3]);
#endif
        }
      }
    }
  }
#endif
  }
#endif
}

#ifndef PNI
#define PNI 10
#endif
#ifndef PWM
#define PWM 2
#endif
#ifndef PWN
#define PWN 2
#endif
#ifndef PMI
#define PMI 2
#endif
constexpr int P_WM = PWM, P_WN = PWN, P_NI = PNI, P_MI = PMI;

static float *g_in0, *g_in1, *g_out1;
static void launch_k(int grid, const float* dA, const float* dB, float* dC, int M, int N, int K) {
  hipLaunchKernelGGL((glds_gemm<P_WM, P_WN, P_NI, P_MI>), dim3(grid), dim3(256), 0, 0, dA, 2 * K, dB, 2 * K, dC, N, M, N, K, g_in0, g_in1, g_out1);
}

static double run(int M, int N, int K, int reps, bool check) {
  std::vector<float> hA((size_t)M * K * 2), hB((size_t)N * K * 2);
  srand(1);
  for (auto& v : hA) v = (float)(rand() & 0xffff) / 65536.0f - 0.5f;
  for (auto& v : hB) v = (float)(rand() & 0xffff) / 65536.0f - 0.5f;
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
  CK(hipMalloc(&g_in0, (size_t)M * N * 4)); CK(hipMalloc(&g_in1, (size_t)M * N * 4)); CK(hipMalloc(&g_out1, (size_t)M * N * 4));
  CK(hipMemset(g_in0, 0, (size_t)M * N * 4)); CK(hipMemset(g_in1, 0, (size_t)M * N * 4));
  const int BM = 16 * P_MI * P_WM;
#ifdef PERSIST
  const int grid = std::min((M + BM - 1) / BM, PERSIST * 256);
#else
  const int grid = (M + BM - 1) / BM;
#endif
  auto launch = [&]() { launch_k(grid, dA, dB, dC, M, N, K); };
  for (int i = 0; i < 30; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double tf = 2.0 * M * N * K / ms / 1e9;
  double maxerr = -1;
#ifdef MIXK
  check = false;
#endif
  if (check && EPI == 0) {
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    maxerr = 0;
    const int rows[] = {0, 1, 15, 16, 31, 32, 63, 64, 65, 1000, M / 2, M - 66, M - 64, M - 2, M - 1};
    for (int r : rows) {
      if (r < 0 || r >= M) continue;
      for (int n = 0; n < N; ++n) {
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)hA[(size_t)r * 2 * K + k] * hB[(size_t)n * 2 * K + k];
        const double e = fabs(s - hC[(size_t)r * N + n]);
        if (!(e <= maxerr)) maxerr = e;        // NaN-propagating
      }
    }
  }
  printf("M=%6d N=%4d K=%5d: %8.4f ms  %7.2f TF (%5.1f%% of 157.3)  maxerr %.3e\n", M, N, K, ms, tf, 100 * tf / 157.3, maxerr);
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(g_in0)); CK(hipFree(g_in1)); CK(hipFree(g_out1));
  return tf;
}

int main(int argc, char** argv) {
#ifdef PERSIST
  printf("PERSIST=%d ", PERSIST);
#endif
#ifdef MIXK
  printf("MIXK=%d ", MIXK);
#endif
  printf("variant: STAGES=%d WAVES_EU=%d EPI=%d ELDS=%d %s\n", STAGES, WAVES_EU, EPI,
#ifdef ELDS
 1,
#else
 0,
#endif
#ifdef NOPIPE
         "NOPIPE"
#else
         "PIPE"
#endif
  );
  if (argc > 3) { run(atoi(argv[2]), atoi(argv[3]), 600, atoi(argv[1]), false); return 0; }
  if (argc > 2) { run(atoi(argv[2]), 300, 600, atoi(argv[1]), false); return 0; }
  if (argc > 1) { run(96000, 300, 600, atoi(argv[1]), false); return 0; }     // long loop for power / clock sampling
#ifdef NHALF
  run(124256, 160, 300, 30, true);
  run(124256, 160, 600, 30, false);
  run(192000, 160, 300, 30, false);
  run(192000, 160, 600, 30, false);
  return 0;
#endif
#ifdef KSWEEP
  run(1000, 300, 300, 2, true);
  run(62128, 300, 300, 30, true);
  run(62128, 300, 600, 30, true);
  run(96000, 300, 300, 30, false);
  run(14208, 300, 600, 30, false);
  run(14208, 300, 300, 30, true);
  run(14208, 300, 900, 30, false);
  return 0;
#endif
  run(1000, 300, 300, 2, true);           // ragged M, K tail
  run(96000, 300, 300, 20, true);
  run(96000, 300, 600, 20, false);
  run(96000, 300, 1200, 20, false);
  run(62000, 300, 600, 20, true);
  run(96000, 320, 1216, 20, false);
  return 0;
}
