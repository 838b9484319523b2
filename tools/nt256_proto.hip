// Standalone prototype of a bf16 NT GEMM K loop for gfx950:  C[M][N] (bf16) = A[M][K] . B[N][K]^T, fp32 accumulation.
// Structure under test: 256 x 256 x 64 tile, 8 waves (2 x 4), one workgroup per CU; the two wave rows run
// STAGGERED by one barrier interval, so that on every SIMD one wave is in its MFMA section while the other one reads its
// fragments out of LDS and issues the LDS-DMA of a later K tile ("ping-pong": a producer / consumer split in time).
// A K tile is four phases of 16 MFMAs (one quadrant of the wave's 128 x 64 output each); LDS holds two K tiles (128 KB)
// in 128-byte rows (full cache lines per row and K tile), 16-byte chunks XOR-swizzled with (row >> 1) & 7 on the source
// side of the DMA; a half-tile's slot is restaged for tile t+2 as soon as both wave rows have read it (prefetch distance
// ~6 phases with two buffers).
// Build: hipcc --offload-arch=gfx950 -O3 -o nt256_proto tools/nt256_proto.hip ; run: ./nt256_proto M N K [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned short u16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#ifndef VMW
#define VMW "10"
#endif
#ifndef NOSTORE
#define NOSTORE 0
#endif
#ifndef SAMEROW
#define SAMEROW 0
#endif
#ifndef PRIO
#define PRIO 1
#endif

template <int OFF> __device__ __forceinline__ f32x4 lds_rd(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

struct Params {
  const u16* A; const u16* B; u16* C;
  int M, N, K, lda, ldb, ldc, m_tiles, n_tiles;
};

__global__ void __launch_bounds__(512, 1) __attribute__((amdgpu_waves_per_eu(2, 2)))
nt256_kernel(const Params P) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr unsigned OOB = 0x80000000u;
  constexpr int BUF = 65536, BOFF = 32768;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, q = lane >> 4;

  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int m_tile = xcd + 8 * (slot / P.n_tiles), n_tile = slot % P.n_tiles;
  if (m_tile >= P.m_tiles) return;
  const int m0 = m_tile * 256, n0 = n_tile * 256;
  const int M = P.M, N = P.N;
  const int T = P.K / 64;

  // ---- DMA slots: wave w stages instructions i = 2w + j (j = 0, 1) of every half-tile; lane L -> (row L >> 3, slot L & 7)
  unsigned a_vo[2][2], b_vo[2][2];
  int a_lds[2][2], b_lds[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7);
      const int arow0 = wr * 128 + h * 64 + (2 * wc + j) * 8;
      const int brow0 = (wave >> 1) * 64 + h * 32 + (2 * (wave & 1) + j) * 8;
      const int gm = m0 + arow0 + (lane >> 3), gn = n0 + brow0 + (lane >> 3);
      a_vo[h][j] = gm < M ? (unsigned)(SAMEROW ? (gm & 255) : gm) * (unsigned)P.lda * 2u + (unsigned)c * 16u : OOB;
      b_vo[h][j] = gn < N ? (unsigned)gn * (unsigned)P.ldb * 2u + (unsigned)c * 16u : OOB;
      a_lds[h][j] = arow0 * 128;
      b_lds[h][j] = BOFF + brow0 * 128;
    }
  const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, 0x7fffffff, 0x00020000);
  const rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, 0x7fffffff, 0x00020000);
  // stage half h of operand (isB) of K tile t into buffer t & 1 (tiles >= T: the out-of-range marker -> zeros, so that
  // the vmcnt bookkeeping stays uniform up to the last tile)
  auto stage = [&](int t, auto ISB, auto H) __attribute__((always_inline)) {
    constexpr bool isB = decltype(ISB)::value;
    constexpr int h = decltype(H)::value;
    const bool ok = t < T;
    const int so = t * 128;
    const int lb = (t & 1) * BUF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (isB)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(smem + lb + b_lds[h][j]), 16,
                                                 ok ? b_vo[h][j] : OOB, so, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(smem + lb + a_lds[h][j]), 16,
                                                 ok ? a_vo[h][j] : OOB, so, 0, 0);
    }
  };
  constexpr std::integral_constant<bool, false> OPA{};
  constexpr std::integral_constant<bool, true> OPB{};
  constexpr std::integral_constant<int, 0> H0{};
  constexpr std::integral_constant<int, 1> H1{};

  // ---- fragment read offsets: lane (l15, q) reads chunk (ks * 4 + q) ^ ((l15 >> 1) & 7) of row l15 of a 16-row tile
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int sw = (q ^ (l15 >> 1)) & 7;
  unsigned a_fo[2][2], b_fo[2][2];      // [buffer][ks]
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_fo[b][ks] = lds0 + (unsigned)(b * BUF + (wr * 128 + l15) * 128 + ((sw ^ (4 * ks)) * 16));
      b_fo[b][ks] = lds0 + (unsigned)(b * BUF + BOFF + (wc * 64 + l15) * 128 + ((sw ^ (4 * ks)) * 16));
    }

  f32x4 acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 aF[4][2], bF[2][2][2];      // A: [mi][ks] of the current row half; B: [col half][ni][ks]

#define RD_A(B_, H_)                                                                  \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                  \
    aF[0][ks] = lds_rd<(H_) * 8192 + 0 * 2048>(a_fo[B_][ks]);                         \
    aF[1][ks] = lds_rd<(H_) * 8192 + 1 * 2048>(a_fo[B_][ks]);                         \
    aF[2][ks] = lds_rd<(H_) * 8192 + 2 * 2048>(a_fo[B_][ks]);                         \
    aF[3][ks] = lds_rd<(H_) * 8192 + 3 * 2048>(a_fo[B_][ks]);                         \
  }
#define RD_B(B_, H_)                                                                  \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                  \
    bF[H_][0][ks] = lds_rd<(H_) * 4096 + 0 * 2048>(b_fo[B_][ks]);                     \
    bF[H_][1][ks] = lds_rd<(H_) * 4096 + 1 * 2048>(b_fo[B_][ks]);                     \
  }
#define QUAD(RH_, CH_)                                                                \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                    \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                  \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                \
        acc[(RH_) * 4 + mi][(CH_) * 2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16( \
            __builtin_bit_cast(bf16x8, bF[CH_][ni][ks]), __builtin_bit_cast(bf16x8, aF[mi][ks]), acc[(RH_) * 4 + mi][(CH_) * 2 + ni], 0, 0, 0);
#define END_L_WAIT()  do { asm volatile("s_waitcnt vmcnt(" VMW ") lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define END_L_NOVM()  do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define M_SECTION(RH_, CH_) do { if (PRIO) __builtin_amdgcn_s_setprio(1); QUAD(RH_, CH_) if (PRIO) __builtin_amdgcn_s_setprio(0); \
    __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

  // ---- prologue: tile 0 entirely, tile 1 except its second A half (phase 1 of tile 0 stages that one)
  stage(0, OPA, H0); stage(0, OPB, H0); stage(0, OPB, H1); stage(0, OPA, H1);
  stage(1, OPA, H0); stage(1, OPB, H0); stage(1, OPB, H1);
#ifdef TWOPHASE
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#else
  asm volatile("s_waitcnt vmcnt(" VMW ")" ::: "memory");
#endif
  asm volatile("s_barrier" ::: "memory");
  if (wr == 1) asm volatile("s_barrier" ::: "memory");      // the second wave row runs one barrier interval behind
  __builtin_amdgcn_sched_barrier(0);

#define TILE(B_)                                                                      \
  /* phase 1 */ RD_B(B_, 0) RD_A(B_, 0) stage(t + 1, OPA, H1); END_L_WAIT(); M_SECTION(0, 0); \
  /* phase 2 */ RD_B(B_, 1) stage(t + 2, OPA, H0); END_L_WAIT(); M_SECTION(0, 1);     \
  /* phase 3 */ RD_A(B_, 1) stage(t + 2, OPB, H0); END_L_NOVM(); M_SECTION(1, 1);     \
  /* phase 4 */ stage(t + 2, OPB, H1); END_L_WAIT(); M_SECTION(1, 0);

#ifdef TWOPHASE
  // two phases per K tile: phase 1 reads A0 + B0 + B1 (16 fragment reads) and computes quadrants (0,0), (0,1); phase 2 reads A1 and
  // computes (1,1), (1,0).  Slots: A0, B0, B1 are free after phase 1 (both wave rows), A1 after phase 2.  Stages per wave and phase: 4
  // DMA instructions (two half-tiles): phase 1 of tile t stages A1(t+1) [free since (t-1, P2)] and ... see TILE2 below.
#define M2_SECTION(RH_, C0_, C1_) do { if (PRIO) __builtin_amdgcn_s_setprio(1); QUAD(RH_, C0_) QUAD(RH_, C1_) if (PRIO) __builtin_amdgcn_s_setprio(0); \
    __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
  // issue order per wave: (t,P1): A1(t+1); (t,P2): A0(t+2), B0(t+2), B1(t+2).  Needed: A0,B0,B1(t+1) at (t+1,P1): issued at (t-1,P2);
  // wait at the end of (t,P2): younger = A1(t+1) [t,P1] + A0,B0,B1(t+2) [t,P2] = 8 instructions -> vmcnt(8).
  // A1(t+1) needed at (t+1,P2): wait at the end of (t+1,P1): younger = A0,B0,B1(t+2), A1(t+2) = 8 -> vmcnt(8).
#define WAIT8()  do { asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TILE2(B_)                                                                     \
  /* phase 1 */ RD_B(B_, 0) RD_B(B_, 1) RD_A(B_, 0) stage(t + 1, OPA, H1); WAIT8(); M2_SECTION(0, 0, 1); \
  /* phase 2 */ RD_A(B_, 1) stage(t + 2, OPA, H0); stage(t + 2, OPB, H0); stage(t + 2, OPB, H1); WAIT8(); M2_SECTION(1, 1, 0);
  int t = 0;
  for (; t + 1 < T; t += 2) {
    TILE2(0)
    ++t;
    TILE2(1)
    --t;
  }
  if (t < T) { TILE2(0) }
#else
  int t = 0;
  for (; t + 1 < T; t += 2) {
    TILE(0)
    ++t;
    TILE(1)
    --t;
  }
  if (t < T) { TILE(0) }
#endif
  if (wr == 0) asm volatile("s_barrier" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- prototype epilogue: fragment-shaped bf16 stores
  if (NOSTORE) {
    float s = 0.f;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) s += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
    if (s == 12345.678f) P.C[0] = 1;
    return;
  }
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int row = m0 + wr * 128 + mi * 16 + l15;
    if (row < M) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int col = n0 + wc * 64 + ni * 16 + 4 * q;
        if (col < N)
          *reinterpret_cast<uint2*>(P.C + (size_t)row * P.ldc + col) =
              make_uint2(pack_bf16(acc[mi][ni][0], acc[mi][ni][1]), pack_bf16(acc[mi][ni][2], acc[mi][ni][3]));
      }
    }
  }
#endif
}

static u16 f2bf(float f) {
  unsigned u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
static float bf2f(u16 h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 62208, N = argc > 2 ? atoi(argv[2]) : 768, K = argc > 3 ? atoi(argv[3]) : 768;
  const int reps = argc > 4 ? atoi(argv[4]) : 20;
  if (K % 64) { printf("K must be a multiple of 64\n"); return 1; }
  std::vector<u16> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
  for (auto& v : hA) v = f2bf(rnd());
  for (auto& v : hB) v = f2bf(rnd() * 0.05f);
  u16 *dA, *dB, *dC;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, hC.size() * 2));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0, hC.size() * 2));
  Params P{dA, dB, dC, M, N, K, K, K, N, (M + 255) / 256, (N + 255) / 256};
  const int grid = 8 * ((P.m_tiles + 7) / 8) * P.n_tiles;
  CK(hipFuncSetAttribute((const void*)nt256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int i = 0; i < 20; ++i) nt256_kernel<<<grid, 512, 131072>>>(P);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0.f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) nt256_kernel<<<grid, 512, 131072>>>(P);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    best = ms < best ? ms : best; sum += ms;
  }
  const double fl = 2.0 * M * N * K;
  printf("nt256 M=%d N=%d K=%d grid %d: best %.1f us (%.0f TF)  mean %.1f us (%.0f TF)\n", M, N, K, grid, best * 1e3, fl / best / 1e9,
         sum / 5 * 1e3, fl / (sum / 5) / 1e9);
  if (!NOSTORE && !SAMEROW) {
    CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0.0; int bad = 0;
    unsigned s2 = 777u;
    for (int it = 0; it < 4000; ++it) {
      s2 = s2 * 1664525u + 1013904223u; const int r = (it < 600) ? (M - 1 - (it % 300)) * (it < 300) + (it % 300) * (it >= 300) : (int)((s2 >> 4) % (unsigned)M);
      s2 = s2 * 1664525u + 1013904223u; const int c = (int)((s2 >> 4) % (unsigned)N);
      double ref = 0.0;
      for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)r * K + k]) * (double)bf2f(hB[(size_t)c * K + k]);
      const double got = bf2f(hC[(size_t)r * N + c]);
      const double err = fabs(got - ref);
      if (err > worst) worst = err;
      if (err > 0.02 + 0.01 * fabs(ref)) { if (bad < 5) printf("  MISMATCH C[%d][%d] = %f, expected %f\n", r, c, got, ref); ++bad; }
    }
    printf("  check: worst abs error %.4g over 4000 samples, %d bad\n", worst, bad);
  }
  return 0;
}
