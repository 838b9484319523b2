// Standalone prototype of the bf16 weight-gradient ("TN") GEMM on the ping-pong loop structure of tools/nt256_proto.hip:
//     Cpart[ks][I][J] (fp32) = sum_{m in chunk ks} A[m][I] . B[m][J]          A, B bf16, k-major as they sit in memory
// 256 x 256 x 64 tile, 8 waves (2 x 4, wave tile 128 x 64 as four 64 x 32 quadrants), one workgroup per CU, the two wave rows one
// barrier interval apart.  What differs from the NT loop is the LDS image and the fragment read:
//   * a half-tile is 64 k-rows x 128 columns (256 B per k-row = two whole cache lines of the operand row); half h of A = output rows
//     i0 + 128 h + [0, 128) (wave row wm owns 64 wm + [0, 64) of them), half h of B = output columns j0 + 128 h + [0, 128) (wave column wn
//     owns 32 wn + [0, 32)) -- the wave's 128 x 64 outputs are NOT contiguous, the halves of the operand rows are;
//   * v_mfma_f32_16x16x32_bf16 wants 8 consecutive k per lane = the transpose of the k-major image: two ds_read_b64_tr_b16 per fragment
//     (tools/tr_probe.hip, gemm_tn.hip.h).  All k-rows of a 256-byte-pitch image start on the same bank: 32-byte units of a k-row are
//     XOR-swizzled with (r & 3) | ((r >> 3) & 1) << 2 on the source side of the LDS-DMA, the eight k-rows {0..3, 8..11} (+4, +16) that one
//     32-lane half of a transpose read touches then sit on eight different bank octets.
// Work items (K chunk, tile) are dealt to the XCDs in runs: XCD x takes items [x per, (x + 1) per) of the chunk-major list, so that the
// workgroups sharing an L2 stream the same rows of A and B.
// Build: hipcc --offload-arch=gfx950 -O3 -o tn256_proto tools/tn256_proto.hip ; run: ./tn256_proto Mrows I J ksplit [reps]
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

#ifndef NOSTORE
#define NOSTORE 0
#endif
#ifndef COLSUM
#define COLSUM 0
#endif

struct Params {
  const u16* A; const u16* B; float* C; float* colsum;
  int I, J, K, lda, ldb, kchunk, ksplit, i_tiles, j_tiles, per;
};

__global__ void __launch_bounds__(512, 1) __attribute__((amdgpu_waves_per_eu(2, 2)))
tn256_kernel(const Params P) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr unsigned OOB = 0x80000000u;
  constexpr int BUF = 65536, BOFF = 32768, HALF = 16384;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, q = lane >> 4;

  const int ntiles = P.i_tiles * P.j_tiles;
  const int bid = blockIdx.x;
  const int w = (bid & 7) * P.per + (bid >> 3);
  if ((bid >> 3) >= P.per || w >= ntiles * P.ksplit) return;
  const int ks = w / ntiles, tile = w % ntiles;
  const int i0 = (tile / P.j_tiles) * 256, j0 = (tile % P.j_tiles) * 256;
  const int kbeg = ks * P.kchunk;
  const int kend = min(P.K, kbeg + P.kchunk);
  const int T = (kend - kbeg + 63) / 64;

  // ---- DMA: a half-tile is 16 instructions of 1 KB (4 k-rows x 256 B); wave w stages i = 2 w + j: k-rows 4 i + (lane >> 4), slot lane & 15
  unsigned a_vo[2][2], b_vo[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 4 * (2 * wave + j) + (lane >> 4);
      const int fz = ((r & 3) << 1) | (((r >> 3) & 1) << 3);
      const int c = (lane & 15) ^ fz;
      const int ci = i0 + h * 128 + 8 * c, cj = j0 + h * 128 + 8 * c;
      a_vo[h][j] = ci < P.I ? ((unsigned)(kbeg + r) * (unsigned)P.lda + (unsigned)ci) * 2u : OOB;
      b_vo[h][j] = cj < P.J ? ((unsigned)(kbeg + r) * (unsigned)P.ldb + (unsigned)cj) * 2u : OOB;
    }
  // (rows >= kend of the last K tile: beyond num_records -> zeros.  The K offset rides in the VECTOR offset: the range check does not
  //  have to see the scalar offset for that)
  const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, kend * P.lda * 2, 0x00020000);
  const rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, kend * P.ldb * 2, 0x00020000);
  const unsigned a_step = 64u * (unsigned)P.lda * 2u, b_step = 64u * (unsigned)P.ldb * 2u;
  auto stage = [&](int t, auto ISB, auto H) __attribute__((always_inline)) {
    constexpr bool isB = decltype(ISB)::value;
    constexpr int h = decltype(H)::value;
    const bool ok = t < T;
    const int lb = (t & 1) * BUF + (isB ? BOFF : 0) + h * HALF + wave * 2048;
    const unsigned ko = (unsigned)t * (isB ? b_step : a_step);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned v0 = isB ? b_vo[h][j] : a_vo[h][j];
      const unsigned vo = (ok && v0 != OOB) ? v0 + ko : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? rB : rA, (__attribute__((address_space(3))) void*)(smem + lb + j * 1024), 16, vo, 0, 0, 0);
    }
  };
  constexpr std::integral_constant<bool, false> OPA{};
  constexpr std::integral_constant<bool, true> OPB{};
  constexpr std::integral_constant<int, 0> H0{};
  constexpr std::integral_constant<int, 1> H1{};

  // ---- transpose reads: lane (p = l15, g = q) points at k-row 8 g + (p >> 2) (+4: second read, +32: second k-step), four columns 4 (p & 3)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int f5 = (l15 >> 2) | ((q & 1) << 2);
  const unsigned rbase = lds0 + (unsigned)((8 * q + (l15 >> 2)) * 256 + 8 * (l15 & 3));
  unsigned a_ad[2][4], b_ad[2][2];      // [buffer][tile of the wave's half]
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) a_ad[b][mi] = rbase + (unsigned)(b * BUF + (((wm * 4 + mi) ^ f5) * 32));
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) b_ad[b][ni] = rbase + (unsigned)(b * BUF + BOFF + (((wn * 2 + ni) ^ f5) * 32));
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint2 aF[4][2][2], bF[2][2][2][2];      // A: [mi][ks][read] of the current row half; B: [col half][ni][ks][read]
  float csum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};      // COLSUM: column sums of A (the bias gradient), tiles mi = wn and wn ^ 2 ... see below

#define TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RD_A(B_, H_)                                                                  \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                  \
    TR(aF[mi][0][0], a_ad[B_][mi], (H_) * 16384 + 0);                                 \
    TR(aF[mi][0][1], a_ad[B_][mi], (H_) * 16384 + 1024);                              \
    TR(aF[mi][1][0], a_ad[B_][mi], (H_) * 16384 + 8192);                              \
    TR(aF[mi][1][1], a_ad[B_][mi], (H_) * 16384 + 8192 + 1024);                       \
  }
#define RD_B(B_, H_)                                                                  \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                  \
    TR(bF[H_][ni][0][0], b_ad[B_][ni], (H_) * 16384 + 0);                             \
    TR(bF[H_][ni][0][1], b_ad[B_][ni], (H_) * 16384 + 1024);                          \
    TR(bF[H_][ni][1][0], b_ad[B_][ni], (H_) * 16384 + 8192);                          \
    TR(bF[H_][ni][1][1], b_ad[B_][ni], (H_) * 16384 + 8192 + 1024);                   \
  }
#define FRAG(F_) __builtin_bit_cast(bf16x8, make_uint4((F_)[0].x, (F_)[0].y, (F_)[1].x, (F_)[1].y))
#define QUAD(RH_, CH_)                                                                \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                    \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                  \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                \
        acc[(RH_) * 4 + mi][(CH_) * 2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16( \
            FRAG(bF[CH_][ni][ks]), FRAG(aF[mi][ks]), acc[(RH_) * 4 + mi][(CH_) * 2 + ni], 0, 0, 0);
  // the transpose reads are invisible to the compiler's counters: the wait lists every destination as an in/out operand, so that no use
  // of a register that has not landed yet can be scheduled above it
#define WAIT_A() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aF[0][0][0]), "+v"(aF[0][0][1]), "+v"(aF[0][1][0]), "+v"(aF[0][1][1]), \
    "+v"(aF[1][0][0]), "+v"(aF[1][0][1]), "+v"(aF[1][1][0]), "+v"(aF[1][1][1]), "+v"(aF[2][0][0]), "+v"(aF[2][0][1]), "+v"(aF[2][1][0]), "+v"(aF[2][1][1]), \
    "+v"(aF[3][0][0]), "+v"(aF[3][0][1]), "+v"(aF[3][1][0]), "+v"(aF[3][1][1]) :: "memory")
#define WAIT_B(H_) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bF[H_][0][0][0]), "+v"(bF[H_][0][0][1]), "+v"(bF[H_][0][1][0]), "+v"(bF[H_][0][1][1]), \
    "+v"(bF[H_][1][0][0]), "+v"(bF[H_][1][0][1]), "+v"(bF[H_][1][1][0]), "+v"(bF[H_][1][1][1]) :: "memory")
#define BAR() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define M_SECTION(RH_, CH_) do { __builtin_amdgcn_s_setprio(1); QUAD(RH_, CH_) __builtin_amdgcn_s_setprio(0); BAR(); } while (0)
  // COLSUM: wave (wm, wn) sums tile mi = wn of each row half out of the fragments it holds anyway (32 unpack-adds per half and K tile).
  // The tile is picked by a wave-uniform BRANCH chain: a run-time index into aF[] put the whole fragment array into scratch (5 x slower).
#define CS1(RH_, MI_) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int rd = 0; rd < 2; ++rd) {               \
    const uint2 v_ = aF[MI_][ks][rd];                                                                                                    \
    csum[RH_][0] += (__builtin_bit_cast(float, v_.x << 16) + __builtin_bit_cast(float, v_.x & 0xffff0000u)) +                            \
                    (__builtin_bit_cast(float, v_.y << 16) + __builtin_bit_cast(float, v_.y & 0xffff0000u)); }
#define CSUM(RH_) do { if (COLSUM) { if (wn == 0) { CS1(RH_, 0) } else if (wn == 1) { CS1(RH_, 1) } else if (wn == 2) { CS1(RH_, 2) } else { CS1(RH_, 3) } } } while (0)

  // ---- prologue: tile 0 entirely, tile 1 except its second A half (phase 1 of tile 0 stages that one)
  stage(0, OPA, H0); stage(0, OPB, H0); stage(0, OPB, H1); stage(0, OPA, H1);
  stage(1, OPA, H0); stage(1, OPB, H0); stage(1, OPB, H1);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  if (wm == 1) asm volatile("s_barrier" ::: "memory");      // the second wave row runs one barrier interval behind
  __builtin_amdgcn_sched_barrier(0);

#define TILE(B_)                                                                                                                   \
  /* phase 1 */ RD_B(B_, 0) RD_A(B_, 0) stage(t + 1, OPA, H1); asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); WAIT_B(0); WAIT_A(); BAR(); M_SECTION(0, 0); \
  /* phase 2 */ RD_B(B_, 1) stage(t + 2, OPA, H0); CSUM(0); asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); WAIT_B(1); BAR(); M_SECTION(0, 1);            \
  /* phase 3 */ RD_A(B_, 1) stage(t + 2, OPB, H0); WAIT_A(); BAR(); M_SECTION(1, 1);                                               \
  /* phase 4 */ stage(t + 2, OPB, H1); CSUM(1); asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); BAR(); M_SECTION(1, 0);

  int t = 0;
  for (; t + 1 < T; t += 2) {
    TILE(0)
    ++t;
    TILE(1)
    --t;
  }
  if (t < T) { TILE(0) }
  if (wm == 0) asm volatile("s_barrier" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (NOSTORE) {
    float s = csum[0][0] + csum[0][1] + csum[1][0] + csum[1][1];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) s += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
    if (s == 12345.678f) P.C[0] = 1;
    return;
  }
  if (COLSUM && (tile % P.j_tiles) == 0) {
    // csum[rh][e]: lane (l15, q) holds k = 8 q .. 8 q + 7 of column l15 of tile mi = wn, summed over the even (e = 0) / odd k
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      float v = csum[rh][0] + csum[rh][1];
      v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      const int c = i0 + rh * 128 + wm * 64 + wn * 16 + l15;
      if (q == 0 && c < P.I) P.colsum[(size_t)ks * P.I + c] = v;
    }
  }
  float* const C = P.C + (size_t)ks * (size_t)P.I * (size_t)P.J;
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int row = i0 + (mi >> 2) * 128 + wm * 64 + (mi & 3) * 16 + l15;
    if (row < P.I) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int col = j0 + (ni >> 1) * 128 + wn * 32 + (ni & 1) * 16 + 4 * q;
        if (col < P.J) __builtin_nontemporal_store(acc[mi][ni], reinterpret_cast<f32x4*>(C + (size_t)row * P.J + col));
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
  const int K = argc > 1 ? atoi(argv[1]) : 62208, I = argc > 2 ? atoi(argv[2]) : 1536, J = argc > 3 ? atoi(argv[3]) : 1536;
  int ksplit = argc > 4 ? atoi(argv[4]) : 0;
  const int reps = argc > 5 ? atoi(argv[5]) : 10;
  const int it = (I + 255) / 256, jt = (J + 255) / 256;
  if (ksplit <= 0) ksplit = 256 / (it * jt) > 0 ? 256 / (it * jt) : 1;
  int kchunk = (K + ksplit - 1) / ksplit;
  kchunk = (kchunk + 63) / 64 * 64;
  ksplit = (K + kchunk - 1) / kchunk;
  const int lda = (I + 7) / 8 * 8, ldb = (J + 7) / 8 * 8;
  std::vector<u16> hA((size_t)K * lda), hB((size_t)K * ldb);
  std::vector<float> hC((size_t)ksplit * I * J), hS((size_t)ksplit * I);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
  for (auto& v : hA) v = f2bf(rnd());
  for (auto& v : hB) v = f2bf(rnd() * 0.05f);
  u16 *dA, *dB; float *dC, *dS;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, hC.size() * 4)); CK(hipMalloc(&dS, hS.size() * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0xff, hC.size() * 4)); CK(hipMemset(dS, 0xff, hS.size() * 4));
  const int total = it * jt * ksplit, per = (total + 7) / 8;
  Params P{dA, dB, dC, dS, I, J, K, lda, ldb, kchunk, ksplit, it, jt, per};
  const int grid = 8 * per;
  CK(hipFuncSetAttribute((const void*)tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int i = 0; i < 5; ++i) tn256_kernel<<<grid, 512, 131072>>>(P);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0.f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) tn256_kernel<<<grid, 512, 131072>>>(P);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    best = ms < best ? ms : best; sum += ms;
  }
  const double fl = 2.0 * K * (double)I * J;
  printf("tn256 K=%d I=%d J=%d ksplit %d (chunk %d) grid %d: best %.1f us (%.0f TF)  mean %.1f us (%.0f TF)\n", K, I, J, ksplit, kchunk, grid,
         best * 1e3, fl / best / 1e9, sum / 5 * 1e3, fl / (sum / 5) / 1e9);
  if (!NOSTORE) {
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hS.data(), dS, hS.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0; int bad = 0;
    unsigned s2 = 777u;
    for (int n = 0; n < 3000; ++n) {
      s2 = s2 * 1664525u + 1013904223u; int r = (int)((s2 >> 4) % (unsigned)I);
      s2 = s2 * 1664525u + 1013904223u; int c = (int)((s2 >> 4) % (unsigned)J);
      if (n < 256) { r = n % I; c = (n * 7) % J; }
      if (n >= 256 && n < 512) { r = I - 1 - (n - 256) % I; c = J - 1 - ((n - 256) * 5) % J; }
      double ref = 0.0;
      for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)k * lda + r]) * (double)bf2f(hB[(size_t)k * ldb + c]);
      double got = 0.0;
      for (int k2 = 0; k2 < ksplit; ++k2) got += hC[((size_t)k2 * I + r) * J + c];
      const double err = fabs(got - ref);
      if (err > worst) worst = err;
      if (!(err <= 0.02 + 0.002 * fabs(ref))) { if (bad < 8) printf("  MISMATCH C[%d][%d] = %f, expected %f\n", r, c, got, ref); ++bad; }
    }
    printf("  check: worst abs error %.4g over 3000 samples, %d bad\n", worst, bad);
    if (COLSUM) {
      int badc = 0; double wc = 0.0;
      for (int r = 0; r < I; r += 3) {
        double ref = 0.0, got = 0.0;
        for (int k = 0; k < K; ++k) ref += bf2f(hA[(size_t)k * lda + r]);
        for (int k2 = 0; k2 < ksplit; ++k2) got += hS[(size_t)k2 * I + r];
        const double err = fabs(got - ref);
        if (err > wc) wc = err;
        if (!(err <= 0.05 + 0.002 * fabs(ref))) { if (badc < 5) printf("  COLSUM MISMATCH [%d] = %f, expected %f\n", r, got, ref); ++badc; }
      }
      printf("  colsum check: worst %.4g, %d bad\n", wc, badc);
    }
  }
  return 0;
}
