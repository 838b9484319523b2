// MEASURED AND DROPPED (round 6, tool build only: GH_PP32_ROWS): 7 % slower than the tiles it was meant to replace -- K loops alone 3.37 against
// 3.15 ms per headline step, whole launches 3.98 against 3.72; two and three LDS buffers alike; the start stagger below: no effect.
// Ping-pong K loop of gemm_nt_kernel<4, 2, 5, 2, 0> (exact fp32, 128 x 160 x 16 tile, 8 waves of 32 x 80, TWO workgroups per CU).
// Included INSIDE the kernel body (gemm_nt.hip.h, `if constexpr (PP32)`).
//
// Why: the 64 x 320 kernel's K loop is MFMA-bound only with three workgroups per CU -- one wave per SIMD and
// workgroup, each with bubbles (barrier per K tile, LDS latency) that the other two cover.  When one workgroup is in its epilogue the
// other two no longer saturate the matrix pipe, so epilogue time ADDS to K-loop time (measured in every round: time ~ FLOPs / 130-140 TF
// + epilogue bytes / 5-6.7 TB/s).  Here a workgroup saturates the pipe BY ITSELF: 8 waves = two per SIMD, the upper two wave rows one
// barrier interval behind the lower two (the structure of gemm_nt_pp.hip.h), so that a SIMD's two waves alternate between a load
// section (7 fragment reads) and an MFMA section (40 x v_mfma_f32_16x16x4_f32 = 1280 matrix cycles).  Two such workgroups per CU
// (<= 128 VGPRs, 51 KB of LDS each): while one is in its epilogue the other one's K loop still runs at the full matrix rate.
//
//   * LDS: THREE K tiles of 18 KB (two buffers = one tile of prefetch distance measured slower than the 64 x 320 kernel's K loop) (A 128 rows x 64 B, B 160 rows x 64 B), the [row][16 k] image of the other fp32 tiles (chunk slot
//     XOR-swizzled with {0,2,3,1}[(row >> 2) & 3] on the source side of the LDS-DMA; one ds_read_b128 = the four k-steps of a 16-row tile).
//   * a K tile is ONE phase: load section (A 2 + B 5 fragment reads of tile t), barrier, MFMA section (40 MFMAs), barrier.
//   * LDS-DMA of tile t + 2 (3 instructions per wave: one of A's eight 16-row pieces, one or two of B's ten -- waves without a second
//     B piece issue it with the out-of-range marker into a scratch KB, so that every wave's count is the same): issued by the lower
//     wave rows at the START of their load section L(t), by the upper rows at the start of their MFMA section M(t-1) -- the SAME
//     barrier interval, the first one in which both halves have read the tile that lived in that buffer (tile t - 1); it has two
//     intervals (~2 us) to land.  Every wave waits for its own pieces (vmcnt(0): nothing else is in flight inside the loop) before
//     the barrier that precedes the first reader: the lower rows at the end of M(t), the upper rows at the end of L(t).
//   K tails (K = 300: 18.75 tiles) and rows beyond M / N: out-of-range marker -> zeros in LDS, as in the other tiles.
{
  constexpr int PSTAGE = BM * 64 + BN * 64;          // 18 432 B
  constexpr int NBUF = 3;                            // K tiles resident in LDS: prefetch distance two tiles (~4 barrier intervals)
  constexpr int PSCRATCH = NBUF * PSTAGE;            // 1 KB that nobody reads
  static_assert(PSCRATCH + 1024 <= SMEM, "scratch KB of the marker DMA");
  const bool upper = wm >= 2;                        // wave rows 2, 3: one barrier interval behind rows 0, 1
  // DMA pieces of this wave: A piece `wave` (16 rows), B pieces `wave` and `wave + 8` (the latter exists for waves 0, 1)
  const int drow_ = lane >> 2;
  const int dkq_ = (lane & 3) ^ nt_swz((lane >> 4) & 3);
  unsigned pa_vo[2], pb_vo[2][2];
  {
    const int gm = m0 + 16 * wave + drow_;
    const int gmc = min(gm, M - 1);
    int s0 = gmc, s1 = gmc;
    if (P.seg[0].gatherA) s0 = P.seg[0].gatherA[gmc];
    if (nseg > 1 && P.seg[1].gatherA) s1 = P.seg[1].gatherA[gmc];
    pa_vo[0] = gm < M ? (unsigned)s0 * (unsigned)lda0 * 4u + (unsigned)dkq_ * 16u : OOB;
    pa_vo[1] = gm < M ? (unsigned)s1 * (unsigned)lda1 * 4u + (unsigned)dkq_ * 16u : OOB;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ib = wave + 8 * j;
      const int n = 16 * ib + drow_;
      const bool ok = ib < BN / 16 && n < N;
      pb_vo[0][j] = ok ? (unsigned)n * (unsigned)ldb0 * 4u + (unsigned)dkq_ * 16u : OOB;
      pb_vo[1][j] = ok ? (unsigned)n * (unsigned)ldb1 * 4u + (unsigned)dkq_ * 16u : OOB;
    }
  }
  auto pp_dma = [&](int t) __attribute__((always_inline)) {
    const int tt = t + toff + tbeg;
    const bool s1 = tt >= nt0;
    const int k0 = s1 ? (tt - nt0) * 16 : tt * 16;
    const int klim = s1 ? K1 : K0;
    const bool kok = t < T && 4 * dkq_ < klim - k0;
    const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(s1 ? A1 : A0), 0, 0x7fffffff, 0x00020000);
    const rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)(s1 ? B1 : B0), 0, 0x7fffffff, 0x00020000);
    unsigned char* sb = smem + (t % NBUF) * PSTAGE;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(sb + wave * 1024), 16,
                                             kok ? (s1 ? pa_vo[1] : pa_vo[0]) : OOB, k0 * 4, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(sb + BM * 64 + wave * 1024), 16,
                                             kok ? (s1 ? pb_vo[1][0] : pb_vo[0][0]) : OOB, k0 * 4, 0, 0);
    // (the second B piece: waves 0, 1 only; the others write zeros into the scratch KB)
    unsigned char* d2 = wave < 2 ? sb + BM * 64 + (wave + 8) * 1024 : smem + PSCRATCH;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)d2, 16,
                                             (kok && wave < 2) ? (s1 ? pb_vo[1][1] : pb_vo[0][1]) : OOB, k0 * 4, 0, 0);
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int fsl_ = q ^ nt_swz((l15 >> 2) & 3);
  const unsigned pa_fo = lds0 + (unsigned)((wrow + l15) * 4 + fsl_) * 16u;
  const unsigned pb_fo = lds0 + (unsigned)(BM * 64) + (unsigned)((wcol + l15) * 4 + fsl_) * 16u;
  f32x4 aF[MI], bF[NI];
#define GH_P32_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define GH_P32_READ(B_)                                                       \
  GH_P32_RD(bF[0], pb_fo, (B_) * PSTAGE + 0 * 1024); GH_P32_RD(bF[1], pb_fo, (B_) * PSTAGE + 1 * 1024);      \
  GH_P32_RD(bF[2], pb_fo, (B_) * PSTAGE + 2 * 1024); GH_P32_RD(bF[3], pb_fo, (B_) * PSTAGE + 3 * 1024);      \
  GH_P32_RD(bF[4], pb_fo, (B_) * PSTAGE + 4 * 1024);                                                         \
  GH_P32_RD(aF[0], pa_fo, (B_) * PSTAGE + 0 * 1024); GH_P32_RD(aF[1], pa_fo, (B_) * PSTAGE + 1 * 1024);
#define GH_P32_BAR() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define GH_P32_MMA()                                                          \
  _Pragma("unroll") for (int s = 0; s < 4; ++s)                               \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                         \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                       \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(bF[ni][s], aF[mi][s], acc[mi][ni], 0, 0, 0);
  // one K tile of buffer B_: lower rows: [DMA(t+1), reads, lgkm] bar [MFMA, vm] bar; upper rows: [reads, lgkm, vm] bar [DMA(t+2), MFMA] bar
#define GH_P32_TILE(B_)                                                       \
  do {                                                                        \
    if (!upper) pp_dma(t + 2);                                                \
    GH_P32_READ(B_)                                                           \
    if (upper) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");    \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   \
    if (drop_mode == 1) drop_a(t, aF);                                        \
    GH_P32_BAR();                                                             \
    if (upper) pp_dma(t + 3);                                                 \
    __builtin_amdgcn_s_setprio(1);                                            \
    GH_P32_MMA()                                                              \
    __builtin_amdgcn_s_setprio(0);                                            \
    if (!upper) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");              \
    GH_P32_BAR();                                                             \
  } while (0)

  // prologue: tiles 0, 1 (everyone); the upper rows also tile 2 (the lower rows issue it in L(0))
  pp_dma(0);
  pp_dma(1);
  // STAGGER.  The two workgroups of a CU are dispatched together and do equal work: left alone they run their K loops together (each
  // at half the matrix rate) and reach their epilogues together -- no overlap, in any round (a finished pair is replaced by a new
  // pair).  The workgroups that the dispatcher places SECOND on each CU (block ids [256, 512) of a launch: it fills the CUs breadth
  // first) therefore start half a K loop late; from then on a slot's next workgroup starts when that slot's epilogue ends, so the
  // offset persists for the whole launch.  The delay costs nothing: the first workgroup has the matrix pipe to itself meanwhile.
  {
    const int units = (dbg_bits >> 8) & 0xff;      // (tool build: GH_DBG = units << 8; 64 cycles per unit and K tile)
    const int stag = units ? units : 10;
    const int pat = (dbg_bits >> 16) & 3;          // which workgroups are "second on their CU": 0 = blocks [256, 512) (breadth-first dispatch),
    const int b_ = (int)blockIdx.x;                //   1 = odd XCD-local index (depth-first), 2 = odd XCD-local index among the first 512, 3 = [512, 1024)
    const bool second = pat == 0 ? (b_ >= 256 && b_ < 512) : pat == 1 ? ((b_ >> 3) & 1) != 0 : pat == 2 ? (b_ < 512 && ((b_ >> 3) & 1) != 0) : (b_ >= 512 && b_ < 1024);
    if (second && stag < 200)
      for (int i = 0; i < T * stag; ++i) __builtin_amdgcn_s_sleep(1);      // 64 cycles each
  }
  if (upper) {
    pp_dma(2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // tile 0's three pieces have landed, the later tiles' may be in flight
  } else {
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  }
  asm volatile("s_barrier" ::: "memory");
  if (upper) asm volatile("s_barrier" ::: "memory");      // the upper wave rows run one barrier interval behind
  __builtin_amdgcn_sched_barrier(0);
  {
    int t = 0;
    for (; t + 2 < T; t += 3) {
      GH_P32_TILE(0);
      ++t;
      GH_P32_TILE(1);
      ++t;
      GH_P32_TILE(2);
      t -= 2;
    }
    if (t < T) { GH_P32_TILE(0); }
    ++t;
    if (t < T) { GH_P32_TILE(1); }
  }
  if (!upper) asm volatile("s_barrier" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (marker pieces of the tiles beyond the last one)
  __syncthreads();                                        // the epilogue stages over the K buffers
#undef GH_P32_RD
#undef GH_P32_READ
#undef GH_P32_BAR
#undef GH_P32_MMA
#undef GH_P32_TILE
}
