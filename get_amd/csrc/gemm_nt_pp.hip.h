// "Ping-pong" K loop of gemm_nt_kernel<2, 4, 4, 8, 2> (bf16 storage, 256 x 256 x 64 tile, 8 waves, one workgroup per CU).
// Included INSIDE the kernel body (gemm_nt.hip.h, `if constexpr (PP)`); prototype with the measurements: tools/nt256_proto.hip.
//
//   * LDS: two K tiles of 64 KB (A 256 rows x 128 B, then B 256 rows x 128 B): a row of a K tile is one full 128-byte line of
//     its operand row (the 64-byte rows of the 32-deep tiles fetch every line twice, one K tile apart -- L1 does not hold it).
//     16-byte chunks XOR-swizzled with (row >> 1) & 7, applied on the SOURCE side of the LDS-DMA (the DMA writes 64 lanes x
//     16 B linearly = 8 rows); a ds_read_b128 fragment read of 16 rows x 4 chunks is bank-conflict free.
//   * A K tile is FOUR phases; a phase is a load section (fragment reads of one quadrant's new operands + the LDS-DMA of
//     one half-tile of a later K tile), a barrier, an MFMA section (16 x v_mfma_f32_16x16x32_bf16 = one 64 x 32 quadrant of
//     the wave's 128 x 64 output over the 64-deep tile), a barrier.  Wave row 1 runs ONE barrier interval behind wave row 0:
//     the two waves of a SIMD (w and w + 4) alternate between the sections, the matrix pipe always has a wave in its MFMA
//     section while the other one's LDS reads and DMA issues run underneath.
//   * Quadrant order (A0,B0) (A0,B1) (A1,B1) (A1,B0): phase 1 reads A0 + B0, phase 2 B1, phase 3 A1, phase 4 nothing.  A
//     half-tile's slot is free once BOTH wave rows have read it (one interval after row 0 did; every load section ends with
//     lgkmcnt(0) before its barrier), and is restaged for tile t + 2 right then: A0 in phase 2, B0 in 3, B1 in 4, A1 in
//     phase 1 of the next tile -- a prefetch distance of ~6 phases with two buffers.
//   * RAW: a wave waits for its own DMA instructions with a COUNTED vmcnt before the barrier that precedes the first reader
//     (never vmcnt(0) inside the loop).  Every stage is two instructions per wave and the issue order is fixed, so "the
//     half-tile read in the next phase has landed" is vmcnt(10) at the end of phases 4, 1 and 2 (five younger stages may be
//     in flight).  Stages beyond the last K tile are issued with the out-of-range marker (zeros into slots nobody reads) so
//     that the count stays uniform.
//   Host-side preconditions (Batch::launch_any): every segment's K a multiple of 64, no K split, no in-loop dropout.
{
  constexpr int PBUF = 65536, PBOFF = 32768;
  // DMA slots: wave w stages instructions i = 2w + j (j = 0, 1) of every half-tile; lane L -> (row L >> 3, slot L & 7).
  // A half h = rows {wm' * 128 + h * 64 + [0, 64)} (what the wave rows read in phase 1 / 3), B half h = columns
  // {wn' * 64 + h * 32 + [0, 32)} (phase 1 / 2).
  unsigned pa_vo[2][2][2], pb_vo[2][2][2];      // [segment][half][j]
  int pa_lds[2][2], pb_lds[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7);
      const int arow0 = wm * 128 + h * 64 + (2 * wn + j) * 8;
      const int brow0 = (wave >> 1) * 64 + h * 32 + (2 * (wave & 1) + j) * 8;
      const int gm = m0 + arow0 + (lane >> 3), gn = brow0 + (lane >> 3);
      const int gmc = min(gm, M - 1);
      int s0 = gmc, s1 = gmc;
      if (P.seg[0].gatherA) s0 = P.seg[0].gatherA[gmc];
      if (nseg > 1 && P.seg[1].gatherA) s1 = P.seg[1].gatherA[gmc];
      pa_vo[0][h][j] = gm < M ? (unsigned)s0 * (unsigned)lda0 * 2u + (unsigned)c * 16u : OOB;
      pa_vo[1][h][j] = gm < M ? (unsigned)s1 * (unsigned)lda1 * 2u + (unsigned)c * 16u : OOB;
      pb_vo[0][h][j] = gn < N ? (unsigned)gn * (unsigned)ldb0 * 2u + (unsigned)c * 16u : OOB;
      pb_vo[1][h][j] = gn < N ? (unsigned)gn * (unsigned)ldb1 * 2u + (unsigned)c * 16u : OOB;
      pa_lds[h][j] = arow0 * 128;
      pb_lds[h][j] = PBOFF + brow0 * 128;
    }
  auto pp_stage = [&](int t, auto ISB, auto H) __attribute__((always_inline)) {
    constexpr bool isB = decltype(ISB)::value;
    constexpr int h = decltype(H)::value;
    const int tt = t + toff + tbeg;
    const bool ok = t < T;
    const bool sg1 = tt >= nt0;
    const int so = (sg1 ? tt - nt0 : tt) * 128;
    const int lb = (t & 1) * PBUF;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(isB ? (sg1 ? B1 : B0) : (sg1 ? A1 : A0)), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned vo = isB ? (sg1 ? pb_vo[1][h][j] : pb_vo[0][h][j]) : (sg1 ? pa_vo[1][h][j] : pa_vo[0][h][j]);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + lb + (isB ? pb_lds[h][j] : pa_lds[h][j])), 16,
                                               ok ? vo : OOB, so, 0, 0);
    }
  };
  constexpr std::integral_constant<bool, false> OPA{};
  constexpr std::integral_constant<bool, true> OPB{};
  constexpr std::integral_constant<int, 0> H0{};
  constexpr std::integral_constant<int, 1> H1{};

  // fragment reads: lane (l15, q) reads chunk (ks * 4 + q) ^ ((l15 >> 1) & 7) of row l15 of a 16-row tile
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int sw = (q ^ (l15 >> 1)) & 7;
  unsigned pa_fo[2][2], pb_fo[2][2];      // [buffer][ks]
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      pa_fo[b][ks] = lds0 + (unsigned)(b * PBUF + (wm * 128 + l15) * 128 + ((sw ^ (4 * ks)) * 16));
      pb_fo[b][ks] = lds0 + (unsigned)(b * PBUF + PBOFF + (wn * 64 + l15) * 128 + ((sw ^ (4 * ks)) * 16));
    }
  f32x4 aF[4][2], bF[2][2][2];      // A: [mi][ks] of the current row half; B: [column half][ni][ks]
  typedef __bf16 pp_bf16x8 __attribute__((ext_vector_type(8)));

#define GH_PP_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define GH_PP_RD_A(B_, H_)                                                            \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                  \
    GH_PP_RD(aF[0][ks], pa_fo[B_][ks], (H_) * 8192 + 0 * 2048);                       \
    GH_PP_RD(aF[1][ks], pa_fo[B_][ks], (H_) * 8192 + 1 * 2048);                       \
    GH_PP_RD(aF[2][ks], pa_fo[B_][ks], (H_) * 8192 + 2 * 2048);                       \
    GH_PP_RD(aF[3][ks], pa_fo[B_][ks], (H_) * 8192 + 3 * 2048);                       \
  }
#define GH_PP_RD_B(B_, H_)                                                            \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                  \
    GH_PP_RD(bF[H_][0][ks], pb_fo[B_][ks], (H_) * 4096 + 0 * 2048);                   \
    GH_PP_RD(bF[H_][1][ks], pb_fo[B_][ks], (H_) * 4096 + 1 * 2048);                   \
  }
#define GH_PP_QUAD(RH_, CH_)                                                          \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                    \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                  \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                \
        acc[(RH_) * 4 + mi][(CH_) * 2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16( \
            __builtin_bit_cast(pp_bf16x8, bF[CH_][ni][ks]), __builtin_bit_cast(pp_bf16x8, aF[mi][ks]), acc[(RH_) * 4 + mi][(CH_) * 2 + ni], 0, 0, 0);
#define GH_PP_END_L(WAIT_)  do { asm volatile(WAIT_ ::: "memory"); __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define GH_PP_M(RH_, CH_) do { __builtin_amdgcn_s_setprio(1); GH_PP_QUAD(RH_, CH_) __builtin_amdgcn_s_setprio(0); \
    __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define GH_PP_TILE(B_)                                                                                                              \
  /* phase 1 */ GH_PP_RD_B(B_, 0) GH_PP_RD_A(B_, 0) pp_stage(t + 1, OPA, H1); GH_PP_END_L("s_waitcnt vmcnt(10) lgkmcnt(0)"); GH_PP_M(0, 0); \
  /* phase 2 */ GH_PP_RD_B(B_, 1) pp_stage(t + 2, OPA, H0); GH_PP_END_L("s_waitcnt vmcnt(10) lgkmcnt(0)"); GH_PP_M(0, 1);           \
  /* phase 3 */ GH_PP_RD_A(B_, 1) pp_stage(t + 2, OPB, H0); GH_PP_END_L("s_waitcnt lgkmcnt(0)"); GH_PP_M(1, 1);                     \
  /* phase 4 */ pp_stage(t + 2, OPB, H1); GH_PP_END_L("s_waitcnt vmcnt(10) lgkmcnt(0)"); GH_PP_M(1, 0);

  // prologue: tile 0 entirely, tile 1 except its second A half (phase 1 of tile 0 stages that one)
  pp_stage(0, OPA, H0); pp_stage(0, OPB, H0); pp_stage(0, OPB, H1); pp_stage(0, OPA, H1);
  pp_stage(1, OPA, H0); pp_stage(1, OPB, H0); pp_stage(1, OPB, H1);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  if (wm == 1) asm volatile("s_barrier" ::: "memory");      // the second wave row runs one barrier interval behind
  __builtin_amdgcn_sched_barrier(0);
  {
    int t = 0;
    for (; t + 1 < T; t += 2) {
      GH_PP_TILE(0)
      ++t;
      GH_PP_TILE(1)
      --t;
    }
    if (t < T) { GH_PP_TILE(0) }
  }
  if (wm == 0) asm volatile("s_barrier" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the marker stages of the last tiles still write (zeros) into the buffers
  __syncthreads();                                        // ... which the epilogue is about to overwrite
#undef GH_PP_RD
#undef GH_PP_RD_A
#undef GH_PP_RD_B
#undef GH_PP_QUAD
#undef GH_PP_END_L
#undef GH_PP_M
#undef GH_PP_TILE
}
