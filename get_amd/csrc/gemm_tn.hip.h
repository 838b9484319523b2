// Grouped fp32 MFMA GEMM, "TN" form (weight gradients), for gfx950:
//     C_p[I][J] (+)= sum_{m in K chunk} A_p[m][I] . B_p[m][J]          (A = upstream gradient rows, B = activation rows)
// split over the M rows ("K" of this product) into L.ksplit chunks; every (chunk, problem, row tile) is one workgroup
// that writes its partial 64 x 320 tile (plain 16-byte stores, summed by reduce_partials_kernel) or adds it atomically.
//
// Same machinery as gemm_nt.hip.h -- operands by LDS-DMA, two LDS stages, one barrier per 16-row K tile -- with the
// operands k-major as they sit in memory ([m][i], [m][j]):
//   * LDS image = the 16 k-rows of the tile back to back, exactly as the DMA delivers them (A: 16 x 256 B, B: 16 x 1280 B;
//     odd k-rows of A are rotated by 128 B through the DMA's source address so that the 8-byte fragment reads of the two
//     k-rows in one LDS access group fall into different bank halves);
//   * INTERLEAVED MFMA tiles: output row tile mi holds rows {wrow + 2 l + mi}, column tile 4g + c holds columns
//     {wcol + 64 g + 4 l + c} (l = 0..15).  A lane's A fragment for all row tiles of a k-step is then ONE 8-byte LDS read
//     and its B fragments for four column tiles ONE 16-byte read: 4 LDS reads per k-step instead of 12, and the four
//     accumulators of a column group are four consecutive columns, i.e. one 16-byte store in the epilogue;
//   * the bias gradient (column sums of A over the chunk) falls out of the A fragments: 2 adds per k-step.
// Preconditions (host-checked, otherwise gemm.hip.h's generic kernel): 16-byte aligned rows, I and J multiples of 4,
// no row gather, 31-bit byte offsets.
#pragma once
#include "gemm.hip.h"

namespace gh {

// BF: opt-in bf16 operand mode (see gemm_nt.hip.h): the four k-steps of a tile are packed into one bf16 MFMA operand
// (slot s of lane q = k-row 4 s + q, for A and B alike), fp32 accumulate.
template <int WM, int WN, int NI, bool BF = false>
__global__ void __launch_bounds__(WM * WN * 64, 3)
gemm_tn_kernel(const Launch L_byval) {
  (void)L_byval;
#if defined(__HIP_DEVICE_COMPILE__)
  const GH_KARG Launch& L = *(const GH_KARG Launch*)__builtin_amdgcn_kernarg_segment_ptr();
  typedef __amdgpu_buffer_rsrc_t rsrc_t;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int MI = 2;
  constexpr int NW = WM * WN, NTHR = NW * 64;
  constexpr int BM = 16 * MI * WM, BN = 16 * NI * WN, BK = 16;
  constexpr int A_BYTES = BK * BM * 4, B_BYTES = BK * BN * 4;
  constexpr int NAI = A_BYTES / 1024, NBI = B_BYTES / 1024;          // DMA instructions per K tile
  constexpr int SA = (NAI + NW - 1) / NW, SB = (NBI + NW - 1) / NW;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int NG4 = NI / 4, NG2 = (NI % 4) / 2;                     // column groups read with 16-byte / 8-byte loads
  static_assert((NI % 2 == 0) && (BM == 64 || BM == 32), "tile shape");
  static_assert(A_BYTES % 1024 == 0 && B_BYTES % 1024 == 0, "whole DMA instructions");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  // ---- work decode: the (row tile, problem) blocks of one K chunk share an XCD (the chunk's rows are read once per L2)
  const int n_inner = L.m_tiles * L.nprob;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int ks = xcd + 8 * (slot / n_inner);
  const int inner = slot % n_inner;
  if (ks >= L.ksplit) return;
  const int prob = inner % L.nprob;
  const int m_tile = inner / L.nprob;
  const GH_KARG Problem& P = L.p[prob];
  const int M = P.M, N = P.N;               // output tile space: M = I rows, N = J columns
  const int m0 = m_tile * BM;
  if (m0 >= M) return;
  const int kbeg = ks * L.kchunk;
  const int kend = min(P.seg[0].K, kbeg + L.kchunk);
  if (kbeg >= kend) return;
  const int T = (kend - kbeg + BK - 1) / BK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int wrow = wm * 16 * MI, wcol = wn * 16 * NI;
  const int l15 = lane & 15, q = lane >> 4;

  const float* A = P.seg[0].A; const float* B = P.seg[0].B;
  const int lda = P.seg[0].lda, ldb = P.seg[0].ldb;
  // rows at or beyond kend read as zeros through the descriptor's range check (the byte offset includes soffset)
  const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, kend * lda * 4, 0x00020000);
  const rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, kend * ldb * 4, 0x00020000);

  // DMA slots: linear chunk c of the A image = (k-row c / (BM/4), physical column chunk c % (BM/4)); odd k-rows hold
  // logical chunk (physical ^ 8): a 128-byte rotation
  unsigned a_vo[SA], b_vo[SB];
#pragma unroll
  for (int j = 0; j < SA; ++j) {
    const int ia = wave + NW * j;
    const int c = ia * 64 + lane;
    const int krow = c / (BM / 4), pc = c % (BM / 4);
    const int lc = (BM == 64) ? (pc ^ ((krow & 1) << 3)) : pc;
    const int col = m0 + 4 * lc;
    a_vo[j] = ((NW * (j + 1) <= NAI || ia < NAI) && col < M) ? ((unsigned)krow * (unsigned)lda + (unsigned)col) * 4u : OOB;
  }
#pragma unroll
  for (int j = 0; j < SB; ++j) {
    const int ib = wave + NW * j;
    const int c = ib * 64 + lane;
    const int krow = c / (BN / 4), pc = c % (BN / 4);
    const int col = 4 * pc;
    b_vo[j] = ((NW * (j + 1) <= NBI || ib < NBI) && col < N) ? ((unsigned)krow * (unsigned)ldb + (unsigned)col) * 4u : OOB;
  }
  auto dma_tile = [&](int t, int st) __attribute__((always_inline)) {
    const int k0 = kbeg + t * BK;
    unsigned char* sb = smem + st * STAGE;
#pragma unroll
    for (int j = 0; j < SA; ++j) {
      const int ia = wave + NW * j;
      if (NW * (j + 1) <= NAI || ia < NAI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(sb + ia * 1024), 16, a_vo[j],
                                                 k0 * lda * 4, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < SB; ++j) {
      const int ib = wave + NW * j;
      if (NW * (j + 1) <= NBI || ib < NBI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(sb + A_BYTES + ib * 1024), 16,
                                                 b_vo[j], k0 * ldb * 4, 0, 0);
    }
  };

  // fragment addresses of k-step s: k-row 4 s + q
  const unsigned a_col = (unsigned)(wrow + 2 * l15);                             // logical float column of this lane's row pair
  const unsigned a_fo0 = (unsigned)q * (BM * 4) + (((BM == 64) ? (a_col ^ ((q & 1) << 5)) : a_col) * 4u);
  const unsigned b_fo = (unsigned)A_BYTES + (unsigned)q * (BN * 4) + (unsigned)(wcol + 4 * l15) * 4u;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2 csum = f32x2{0.f, 0.f};
  float* const colsum = P.colsum;

  struct Frag { f32x2 a; f32x4 b4[NG4 > 0 ? NG4 : 1]; f32x2 b2[NG2 > 0 ? NG2 : 1]; };
  auto read_frag = [&](int st, int s, Frag& f) __attribute__((always_inline)) {
    const unsigned char* sb = smem + st * STAGE + s * 4 * (BM * 4);
    f.a = *reinterpret_cast<const f32x2*>(sb + a_fo0);
    const unsigned char* bb = smem + st * STAGE + s * 4 * (BN * 4) + b_fo;
#pragma unroll
    for (int g = 0; g < NG4; ++g) f.b4[g] = *reinterpret_cast<const f32x4*>(bb + g * 256);
#pragma unroll
    for (int g = 0; g < NG2; ++g) f.b2[g] = *reinterpret_cast<const f32x2*>(smem + st * STAGE + A_BYTES + (s * 4 + q) * (BN * 4) +
                                                                             (unsigned)(wcol + 64 * NG4 + 32 * g + 2 * l15) * 4u);
  };
  // acc[mi][ni][r] = C[i = wrow + 2 (4 q + r) + mi][j = column of tile ni at lane l15]
  auto mma = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NG4; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[mi][4 * g + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[mi], f.b4[g][c], acc[mi][4 * g + c], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < NG2; ++g)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[mi][4 * NG4 + 2 * g + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[mi], f.b2[g][c], acc[mi][4 * NG4 + 2 * g + c], 0, 0, 0);
    if (colsum) csum += f.a;
  };

  dma_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (BF) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    auto pk = [](float a, float b, float c, float d) __attribute__((always_inline)) {
      return __builtin_bit_cast(s16x4, make_uint2(nt_pack_bf16(a, b), nt_pack_bf16(c, d)));
    };
    Frag f[4];
    for (int t = 0; t < T; ++t) {
      const int st = t & 1;
      if (t + 1 < T) dma_tile(t + 1, st ^ 1);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) read_frag(st, s4, f[s4]);
      s16x4 ab[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) ab[mi] = pk(f[0].a[mi], f[1].a[mi], f[2].a[mi], f[3].a[mi]);
      if (colsum) csum += f[0].a + f[1].a + f[2].a + f[3].a;
#pragma unroll
      for (int g = 0; g < NG4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const s16x4 bb = pk(f[0].b4[g][c], f[1].b4[g][c], f[2].b4[g][c], f[3].b4[g][c]);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[mi][4 * g + c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab[mi], bb, acc[mi][4 * g + c], 0, 0, 0);
        }
#pragma unroll
      for (int g = 0; g < NG2; ++g)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const s16x4 bb = pk(f[0].b2[g][c], f[1].b2[g][c], f[2].b2[g][c], f[3].b2[g][c]);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[mi][4 * NG4 + 2 * g + c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab[mi], bb, acc[mi][4 * NG4 + 2 * g + c], 0, 0, 0);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
  Frag f0, f1;
  for (int t = 0; t < T; ++t) {
    const int st = t & 1;
    if (t + 1 < T) dma_tile(t + 1, st ^ 1);
    read_frag(st, 0, f0);
    read_frag(st, 1, f1);
    __builtin_amdgcn_sched_barrier(0);
    mma(f0);
    read_frag(st, 2, f0);
    __builtin_amdgcn_sched_barrier(0);
    mma(f1);
    read_frag(st, 3, f1);
    __builtin_amdgcn_sched_barrier(0);
    mma(f0);
    mma(f1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  }

  if (colsum && wn == 0) {
    // lanes with the same l15 hold the sums of k-rows = q (mod 4): fold the four, lane q = 0 writes the column pair
    csum[0] += __shfl_xor(csum[0], 16); csum[1] += __shfl_xor(csum[1], 16);
    csum[0] += __shfl_xor(csum[0], 32); csum[1] += __shfl_xor(csum[1], 32);
    const int c = m0 + wrow + 2 * l15;
    if (q == 0 && c < M) *reinterpret_cast<f32x2*>(colsum + (size_t)ks * (size_t)P.colsum_stride + c) = csum;
  }

  // ---- epilogue: partial tile (or atomic add).  Row i = m0 + wrow + 2 (4 q + r) + mi; a column group is 4 consecutive j.
  const int ldc = P.ldc;
  float* const C = P.C + (size_t)ks * (size_t)P.split_stride;
  const bool atomic = P.epi == EPI_ATOMIC;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wrow + 2 * (4 * q + r) + mi;
      if (row >= M) continue;
      float* crow = C + (size_t)row * ldc;
#pragma unroll
      for (int g = 0; g < NG4; ++g) {
        const int col = wcol + 64 * g + 4 * l15;
        if (col < N) {
          const float4 v = make_float4(acc[mi][4 * g + 0][r], acc[mi][4 * g + 1][r], acc[mi][4 * g + 2][r], acc[mi][4 * g + 3][r]);
          if (atomic) { atomicAdd(crow + col, v.x); atomicAdd(crow + col + 1, v.y); atomicAdd(crow + col + 2, v.z); atomicAdd(crow + col + 3, v.w); }
          else *reinterpret_cast<float4*>(crow + col) = v;
        }
      }
#pragma unroll
      for (int g = 0; g < NG2; ++g) {
        const int col = wcol + 64 * NG4 + 32 * g + 2 * l15;
        if (col < N) {
          const float2 v = make_float2(acc[mi][4 * NG4 + 2 * g][r], acc[mi][4 * NG4 + 2 * g + 1][r]);
          if (atomic) { atomicAdd(crow + col, v.x); atomicAdd(crow + col + 1, v.y); }
          else *reinterpret_cast<float2*>(crow + col) = v;
        }
      }
    }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------
// bf16 STORAGE variant (BASELINE configs[4]): A [m][I] and B [m][J] hold bf16.  The K tile is 32 rows; the LDS image is
// again the k-major rows as the DMA delivers them (A: 32 x 128 B, B: 32 x 640 B -- the same byte counts as the fp32 tile).
// v_mfma_f32_16x16x32_bf16 wants 8 consecutive k per lane, i.e. the transpose of a k-major image: ds_read_b64_tr_b16
// delivers it.  Measured semantics (tools/tr_probe.hip): within a 16-lane group, lane p reads 8 bytes at ITS address and
// output lane i receives element (i & 3) of lanes (i >> 2) + 4 j, j = 0..3.  With lane p pointing at
// X[k0 + (p >> 2)][c0 + 4 (p & 3) ..+3] the group reads a 4(k) x 16(c) block and lane i gets X[k0..k0+3][c0 + i]: four
// consecutive k of ITS column.  Two such reads (k0 = 8 g, 8 g + 4 for lane group g) make one MFMA operand.
// (Round 5, measured and removed: THREE stages -- prefetch distance 2 on 72 KB of dynamic LDS, vmcnt-counted waits, bare barriers, two
//  workgroups per CU instead of three -- because a K tile is only 20 MFMAs = 320 matrix cycles per wave and PMC had the waves parked
//  41 % of their cycles: gemm_big_tn 1.75 -> 2.01 ms per step on configs[4].  As for the fp32 tiles, the third resident workgroup is
//  worth more than the deeper prefetch.)
// MI = 4 (a 128 x 320 tile, two workgroups per CU): 28 instead of 24 LDS-DMA instructions for twice the MFMAs of a K tile.
template <int WM, int WN, int NI, int MI = 2>
__global__ void __launch_bounds__(WM * WN * 64, MI == 4 ? 2 : 3)
gemm_tn_bf16_kernel(const Launch L_byval) {
  (void)L_byval;
#if defined(__HIP_DEVICE_COMPILE__)
  const GH_KARG Launch& L = *(const GH_KARG Launch*)__builtin_amdgcn_kernarg_segment_ptr();
  typedef __amdgpu_buffer_rsrc_t rsrc_t;
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  constexpr int NW = WM * WN;
  constexpr int BM = 16 * MI * WM, BN = 16 * NI * WN, BK = 32;
  constexpr int A_BYTES = BK * BM * 2, B_BYTES = BK * BN * 2;
  constexpr int NAI = A_BYTES / 1024, NBI = B_BYTES / 1024;
  constexpr int SA = (NAI + NW - 1) / NW, SB = (NBI + NW - 1) / NW;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(A_BYTES % 1024 == 0 && B_BYTES % 1024 == 0, "whole DMA instructions");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int n_inner = L.m_tiles * L.nprob;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int ks = xcd + 8 * (slot / n_inner);
  const int inner = slot % n_inner;
  if (ks >= L.ksplit) return;
  const int prob = inner % L.nprob;
  const int m_tile = inner / L.nprob;
  const GH_KARG Problem& P = L.p[prob];
  const int M = P.M, N = P.N;
  const int m0 = m_tile * BM;
  if (m0 >= M) return;
  const int kbeg = ks * L.kchunk;
  const int kend = min(P.seg[0].K, kbeg + L.kchunk);
  if (kbeg >= kend) return;
  const int T = (kend - kbeg + BK - 1) / BK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int wrow = wm * 16 * MI, wcol = wn * 16 * NI;
  const int l15 = lane & 15, q = lane >> 4;

  const float* A = P.seg[0].A; const float* B = P.seg[0].B;          // bf16 data behind float-typed descriptor fields
  const int lda = P.seg[0].lda, ldb = P.seg[0].ldb;
  const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, kend * lda * 2, 0x00020000);
  const rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, kend * ldb * 2, 0x00020000);

  // DMA: linear 16-byte chunk c of the A image = k-row c / (BM/8), slot c % (BM/8).
  // Bank swizzle (round 5): a 32-lane group of ds_read_b64_tr_b16 reads two 16-byte chunks of each of the k-rows
  // {0,1,2,3,8,9,10,11} (+4 for the second read, +16 for the other group); with the natural row pitches (128 B and 640 B, both
  // = 128 mod 256) rows {0,2,8,10} and {1,3,9,11} each fall on the same 8 banks: a 4-way conflict on every read (PMC on
  // configs[4]: SQ_LDS_BANK_CONFLICT = 75 % of SQ_LDS_IDX_ACTIVE, LDS issue stalls 25 % of the wave cycles).  The DMA writes LDS
  // linearly, so the fix is on the SOURCE side, as in the NT kernel: slot s of k-row r holds logical column chunk s ^ f(r),
  // f(r) = 2 ((r >> 1) & 1) + 4 ((r >> 3) & 1) -- the four rows that share a bank range then sit at four different chunk pairs.
  auto fswz = [](int r) { return (((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2); };
  // (the A image of the 128-row tile has a 256-byte pitch: ALL its k-rows start on the same bank, so the eight rows of a read
  //  need eight different chunk pairs -- three swizzle bits: r & 3 and bit 3 of r)
  auto fswzA = [&](int r) { return BM == 128 ? (((r & 3) << 1) | (((r >> 3) & 1) << 3)) : fswz(r); };
  unsigned a_vo[SA], b_vo[SB];
#pragma unroll
  for (int j = 0; j < SA; ++j) {
    const int ia = wave + NW * j;
    const int c = ia * 64 + lane;
    const int krow = c / (BM / 8), col = m0 + 8 * ((c % (BM / 8)) ^ fswzA(krow));
    a_vo[j] = ((NW * (j + 1) <= NAI || ia < NAI) && col < M) ? ((unsigned)krow * (unsigned)lda + (unsigned)col) * 2u : OOB;
  }
#pragma unroll
  for (int j = 0; j < SB; ++j) {
    const int ib = wave + NW * j;
    const int c = ib * 64 + lane;
    const int krow = c / (BN / 8), col = 8 * ((c % (BN / 8)) ^ fswz(krow));
    b_vo[j] = ((NW * (j + 1) <= NBI || ib < NBI) && col < N) ? ((unsigned)krow * (unsigned)ldb + (unsigned)col) * 2u : OOB;
  }
  auto dma_tile = [&](int t, int st) __attribute__((always_inline)) {
    const int k0 = kbeg + t * BK;
    unsigned char* sb = smem + st * STAGE;
#pragma unroll
    for (int j = 0; j < SA; ++j) {
      const int ia = wave + NW * j;
      if (NW * (j + 1) <= NAI || ia < NAI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(sb + ia * 1024), 16, a_vo[j],
                                                 k0 * lda * 2, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < SB; ++j) {
      const int ib = wave + NW * j;
      if (NW * (j + 1) <= NBI || ib < NBI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(sb + A_BYTES + ib * 1024), 16,
                                                 b_vo[j], k0 * ldb * 2, 0, 0);
    }
  };

  // transpose-read addresses: lane (p = l15, g = q) points at k-row 8 g + (p >> 2) (+4 for the second read), 4 columns at 4 (p & 3)
  // (the swizzle XORs bits 5-6 of the byte offset inside a row; the k-row of the second read, +4, has the same f)
  static_assert((BM == 64 || BM == 128) && BN % 64 == 0, "column chunks are swizzled inside aligned groups of eight (sixteen: BM = 128)");
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int trow = 8 * q + (l15 >> 2);
  const unsigned fz = (unsigned)fswz(trow) << 4, fzA = (unsigned)fswzA(trow) << 4;
  unsigned a_ad[MI], b_ad[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
    a_ad[mi] = lds0 + (unsigned)trow * (BM * 2) + (((unsigned)(wrow + 16 * mi + 4 * (l15 & 3)) * 2u) ^ fzA);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
    b_ad[ni] = lds0 + (unsigned)A_BYTES + (unsigned)trow * (BN * 2) + (((unsigned)(wcol + 16 * ni + 4 * (l15 & 3)) * 2u) ^ fz);
  auto tr_read = [&](unsigned addr) __attribute__((always_inline)) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  float csum[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) csum[mi] = 0.f;
  float* const colsum = P.colsum;

  dma_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const int st = t & 1;
    if (t + 1 < T) dma_tile(t + 1, st ^ 1);
    const unsigned so = (unsigned)st * STAGE;
    uint2 alo[MI], ahi[MI], blo[NI], bhi[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) { alo[mi] = tr_read(a_ad[mi] + so); ahi[mi] = tr_read(a_ad[mi] + so + 4 * (BM * 2)); }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { blo[ni] = tr_read(b_ad[ni] + so); bhi[ni] = tr_read(b_ad[ni] + so + 4 * (BN * 2)); }
    // the reads are invisible to the compiler's counters: wait here, and make every destination an in/out operand of the
    // wait so that no copy of a not-yet-landed register can be scheduled above it (cdna_hip_programming.md, inline-asm form ii)
    static_assert((MI == 2 || MI == 4) && NI == 10, "operand lists below");
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(alo[0]), "+v"(alo[1]), "+v"(ahi[0]), "+v"(ahi[1]), "+v"(blo[0]), "+v"(blo[1]), "+v"(blo[2]), "+v"(blo[3]),
                   "+v"(blo[4]), "+v"(blo[5]), "+v"(blo[6]), "+v"(blo[7]), "+v"(blo[8]), "+v"(blo[9])
                 :: "memory");
    if constexpr (MI == 4)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(alo[2]), "+v"(alo[3]), "+v"(ahi[2]), "+v"(ahi[3]) :: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(bhi[0]), "+v"(bhi[1]), "+v"(bhi[2]), "+v"(bhi[3]), "+v"(bhi[4]), "+v"(bhi[5]), "+v"(bhi[6]), "+v"(bhi[7]),
                   "+v"(bhi[8]), "+v"(bhi[9])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    uint4 af[MI], bfr[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[mi] = make_uint4(alo[mi].x, alo[mi].y, ahi[mi].x, ahi[mi].y);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bfr[ni] = make_uint4(blo[ni].x, blo[ni].y, bhi[ni].x, bhi[ni].y);
    // swapped operands: acc[mi][ni][r] = C[i = wrow + 16 mi + l15][j = wcol + 16 ni + 4 q + r]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[ni]), __builtin_bit_cast(bf16x8, af[mi]),
                                                              acc[mi][ni], 0, 0, 0);
    if (colsum) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const unsigned w[4] = {af[mi].x, af[mi].y, af[mi].z, af[mi].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[mi] += __builtin_bit_cast(float, w[e] << 16) + __builtin_bit_cast(float, w[e] & 0xffff0000u);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (colsum && wn == 0) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      float v = csum[mi];
      v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      const int c = m0 + wrow + 16 * mi + l15;
      if (q == 0 && c < M) colsum[(size_t)ks * (size_t)P.colsum_stride + c] = v;
    }
  }
  const int ldc = P.ldc;
  float* const C = P.C + (size_t)ks * (size_t)P.split_stride;
  const bool atomic = P.epi == EPI_ATOMIC;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int row = m0 + wrow + 16 * mi + l15;
    if (row >= M) continue;
    float* crow = C + (size_t)row * ldc;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = wcol + 16 * ni + 4 * q;
      if (col < N) {
        const float4 v = make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
        if (atomic) { atomicAdd(crow + col, v.x); atomicAdd(crow + col + 1, v.y); atomicAdd(crow + col + 2, v.z); atomicAdd(crow + col + 3, v.w); }
        else *reinterpret_cast<float4*>(crow + col) = v;
      }
    }
  }
#endif
}

}  // namespace gh
