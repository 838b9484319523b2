// Grouped fp32 MFMA GEMM, "NT" form, for gfx950:   C_p[M][N_p] = epilogue_p( sum_seg A_seg[M][K_seg] . B_seg[N_p][K_seg]^T )
// Same problem descriptors as gemm.hip.h; BOTH operands are contraction-contiguous (A row-major activations,
// B = the weight as PyTorch stores it, [n_out][n_in]; the backward's dX = G.W takes the cached transpose).
//
// Preconditions (checked by the host, otherwise the generic kernel in gemm.hip.h runs): every operand row 16-byte
// aligned, K a multiple of 4, byte offsets below 2^31.
//
// Design (measured against the register-staged 64x320 kernel it replaces, tools/glds_proto.hip):
//   * operands go HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write pass, no
//     address VALU -- a K tile costs a wave 6 DMA instructions + a dozen SALU;
//   * LDS image per operand is [row][16 k]: 64-byte rows of four 16-byte chunks, chunk slot XOR-swizzled with
//     {0,2,3,1}[(row>>2)&3] so that a ds_read_b128 of 16 rows x 4 k-quads is bank-conflict free.  The DMA writes
//     LDS linearly (wave base + lane*16), so the swizzle is applied on the SOURCE side: lane L fetches the chunk
//     that belongs at linear position L;
//   * one ds_read_b128 = the fragment of ALL four k-steps of the K tile for one 16-row MFMA tile (k = 4q + s is
//     lane q's element for step s), i.e. 12 LDS reads per K tile and wave instead of 48;
//   * fragments are software-pipelined across the per-tile barrier: the last half of tile t-1's MFMAs is issued
//     after the barrier, underneath the LDS reads of tile t;
//   * epilogues run on whole rows: the accumulator tile is staged through LDS (two passes of 16 rows per wave row)
//     and every epilogue stream (gate inputs, outputs) is read/written as contiguous runs of a row -- fragment-shaped
//     epilogue accesses (16 rows x 64 B per instruction) ran the same streams at 3.8 TB/s, whole rows at 6.7 TB/s;
//     row reductions (attention head scores, the GSL scorer's projection) re-read the finished rows from LDS.
#pragma once
#include "common.h"
#include "gemm.hip.h"

namespace gh {
#ifdef GH_MEASURE
// tool build: s_memtime ticks (10 ns) of the 256-tile epilogue, summed over the workgroups of every launch since the last reset (thread 0):
// [kind][0] = staging (accumulators -> LDS between two barriers), [1] = input requests + compute + stores, [2] = passes, [3] = K loop
__device__ unsigned long long g_nt_phase[16 * 4];
#endif

__device__ __forceinline__ int nt_swz(int j) { return (0x78 >> (2 * j)) & 3; }     // {0,2,3,1}
__device__ __forceinline__ unsigned nt_pack_bf16(float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));     // v_cvt_pk_bf16_f32 (RNE)
}

// MODE 1: opt-in operand mode: fp32 storage, fragments rounded to bf16 in registers (v_cvt_pk_bf16_f32, RNE), one
//         v_mfma_f32_16x16x16_bf16 per K tile (a lane's four k values = its float4); epilogues fp32.
// MODE 2: bf16 STORAGE (BASELINE configs[4]): A and B are bf16 in HBM, the K tile is 32 deep -- the LDS image is the same
//         64-byte rows ([row][32 k bf16]), one ds_read_b128 is a lane's 8 consecutive k, i.e. exactly the operand of
//         v_mfma_f32_16x16x32_bf16 (one MFMA per 16x16 tile and K tile) -- fp32 accumulation and epilogue arithmetic,
//         epilogue streams bf16 or fp32 per Problem::io.
// MI = 4 (bf16 storage only: <2, 2, 8, 4, 2> = a 128 x 256 tile): at h = 768 the 64 x 320 tile wastes a fifth of its
//         columns (768 = 2.4 x 320) and re-reads the 320-row weight panel from L2 for every 64 rows; 128 x 256 covers 768
//         exactly, halves the weight traffic per FLOP and needs 12 instead of 24 ds_read_b128 per 32 MFMAs.  128
//         accumulator registers per lane -> two workgroups per CU.
// MODE 3: fp32 storage and fp32 results on the bf16 MFMA ("fp32x3", gh_set_gemm_mode(2)).  fp32 MFMA runs at 1/16 of the bf16
//         rate on gfx950, so an fp32 product is formed from bf16 pieces instead: every operand value x is split in registers
//         into x = hi + mid + lo, three bf16 numbers (round-to-nearest at each level, the remainders are exact fp32
//         subtractions, |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|), and a . b is accumulated in fp32 from the six largest of the
//         nine cross products: ah.bh + am.bh + al.bh + ah.bm + am.bm + ah.bl.  The dropped terms (am.bl, al.bm, al.bl) are
//         below 2^-23 |a.b| -- the size of ONE fp32 rounding of the product -- and the sums are fp32 adds as in the exact
//         mode.  v_mfma_f32_16x16x32_bf16 contracts 32 k-slots per instruction; a K tile has 16 real k, so each instruction
//         carries TWO of the six terms (slots 0-3 of a lane: one piece of its four k values, slots 4-7: another piece):
//         three MFMAs of 16 cycles per 16 x 16 x 16 product instead of four fp32 MFMAs of 32 cycles.  Measured: in registers
//         (tools/split_probe.hip) 260 fp32-equivalent TFLOP/s against 154 for the fp32 MFMA; in THIS kernel only +3-5 %
//         (M = 96 000, N = 300: 101.7 / 113.9 / 131.9 TF at K = 300 / 600 / 1200 against 97.5 / 110.0 / 127.8 exact, same
//         error against fp64): the ~350 VALU instructions per K tile that split the 12 fragments (48 floats per lane, most of
//         them the WEIGHT fragments every workgroup splits again) take as long as the MFMAs they replace.  The mode stays
//         opt-in and experimental.  Upper bound with the weight pieces pre-split once per optimiser step (measured by feeding the
//         raw fragment bits as "pieces", results invalid): 130.8 / 151.0 / 184.3 TF at K = 300 / 600 / 1200.
template <int WM, int WN, int NI, int MI = 2, int MODE = 0>
// (second argument = waves per SIMD the register budget must allow: 256-thread workgroups put one wave on every SIMD, so it is also
//  the workgroups per CU; the 512-thread 128 x 256 tile puts two, and wants two workgroups = four waves per SIMD)
__global__ void __launch_bounds__(WM * WN * 64, (MI == 8) ? 2 : (WM == 4 && WN == 2 && NI == 5 && MI == 2) ? 4 : (MI == 4 && NI == 4 && WN == 4) ? 4 : (MI == 4 && NI == 4) ? 3 : (MI == 4 || MODE == 3 || MODE == 4) ? 2 : (WM == 2 && WN == 2 && NI == 5 && MODE == 0) ? 4 : 3)
gemm_nt_kernel(const Launch L_byval) {
  (void)L_byval;
#if defined(__HIP_DEVICE_COMPILE__)
  const GH_KARG Launch& L = *(const GH_KARG Launch*)__builtin_amdgcn_kernarg_segment_ptr();
  typedef __amdgpu_buffer_rsrc_t rsrc_t;
  constexpr int NW = WM * WN, NTHR = NW * 64;
  constexpr bool BF = MODE == 1;
  constexpr int ESZ = MODE == 2 ? 2 : 4, KQ = 16 / ESZ;          // element size, elements per 16-byte chunk
  // MI = 8 (bf16 storage only: <2, 4, 4, 8, 2> = a 256 x 256 x 64 tile on 8 waves, one workgroup per CU): the "ping-pong" K loop
  // below (PP) -- the two wave rows run staggered by one barrier interval, so that each SIMD always has one wave in its
  // MFMA section and one reading fragments / issuing LDS-DMA.
  constexpr bool PP = MI == 8;
  constexpr bool PP32_ = MODE == 0 && WM == 4 && WN == 2 && NI == 5 && MI == 2;
  // <4, 2, 5, 2, 0>: the exact-fp32 128 x 160 x 16 tile on 8 waves, two workgroups per CU, ping-pong K loop (gemm_nt_pp32.hip.h)
  constexpr bool PP32 = MODE == 0 && WM == 4 && WN == 2 && NI == 5 && MI == 2;
  constexpr int BM = 16 * MI * WM, BN = 16 * NI * WN, BK = PP ? 64 : 4 * KQ;
  // MODE 4 ("fp32x3" with PRE-SPLIT weights): B points at the weight's split image (split3_kernel: per row and group
  // of four k the three bf16 pieces hi[4] | mid[4] | lo[4], 24 bytes; row pitch a multiple of 16 bytes, passed as ldb in
  // 4-byte units).  A K tile of B is 96 bytes per row; its LDS image has a pitch of 7 chunks = 112 bytes (the seventh chunk
  // is never written by real data) so that the 8-byte piece reads of 16 rows x 2 k-quads fall into distinct banks.
  constexpr bool X3P = MODE == 4;
  constexpr int BROW = X3P ? 112 : 64;                             // bytes of one B row per K tile in LDS
  constexpr int NAI = BM / 16, NBI = X3P ? (BN * 7 + 63) / 64 : BN / 16;   // DMA instructions per K tile (1 KiB each)
  constexpr int SA = (NAI + NW - 1) / NW, SB = (NBI + NW - 1) / NW; // ... per wave
  constexpr int STAGE = BM * 64 + BN * BROW;
  constexpr int EP_PITCH = BN + 4;                                 // floats: 16-byte rows, 8 consecutive rows cover all banks
  constexpr int EP_BYTES = 16 * WM * EP_PITCH * 4 + NW * 64 * 16 + BN * 4;  // staged rows + row-reduction partials + bias
  // K-loop LDS stages.  fp32: a K tile is ~2560 MFMA cycles per wave (~1 us), about one DMA round trip, so two stages
  // (prefetch distance 1) suffice.  The 128 x 256 bf16 tile computes a K tile in 512 cycles (~0.2 us): with distance 1 every
  // tile waits most of a DMA latency (PMC: waves parked 55 % of the time), so it runs three stages / distance 2, waits
  // with vmcnt(DMA instructions of ONE tile) and synchronises with a bare s_barrier (a __syncthreads() would drain the
  // newest tile's DMA as well).  73.5 KB of LDS -> dynamic allocation.
  // (fp32x3, MODE 3, measured with three stages as well: slower -- 122 vs 132 TF at K = 1200 -- its K loop is bound by the
  // VALU work of the in-register splits, not by DMA latency)
  constexpr int NST = (MI == 4 && WM == 4) ? 4 : (MI == 4) ? 3 : 2;      // (256 x 256 / 8 waves: one workgroup per CU, four stages)
  constexpr bool DYN_LDS = NST != 2 || X3P || PP;                        // more than 64 KB: dynamic allocation
  constexpr int SMEM = PP ? 131072 : PP32_ ? (3 * STAGE + 1024 > EP_BYTES ? 3 * STAGE + 1024 : EP_BYTES) : (NST * STAGE > EP_BYTES ? NST * STAGE : EP_BYTES);
  static_assert(!PP || (MODE == 2 && WM == 2 && WN == 4 && NI == 4 && EP_BYTES <= 131072), "ping-pong loop: 256 x 256 bf16 tile on 2 x 4 waves");
  constexpr unsigned OOB = 0x80000000u;
  constexpr int NH = NI / 2;                                       // B fragment batches: X = tiles [0,NH), Y = [NH,NI)
  static_assert(MI == 2 || ((MI == 4 || MI == 8) && MODE == 2), "two 16-row tiles per wave (four / eight in the 128 x 256 / 256 x 256 bf16 configurations)");
  extern __shared__ __attribute__((aligned(16))) unsigned char nt_dyn_smem[];
  __shared__ __attribute__((aligned(16))) unsigned char nt_static_smem[DYN_LDS ? 16 : SMEM];
  unsigned char* const smem = DYN_LDS ? nt_dyn_smem : nt_static_smem;

  // ---- XCD-aware work decode (as gemm.hip.h): the problems of one row tile run back to back on one XCD
  const int n_inner = L.nprob * L.ksplit;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  // (few-row launches -- fewer than 8 row tiles, e.g. the head's 32-row split-K products: the grid is m_tiles x n_inner and
  //  consecutive workgroups, i.e. the 8 XCDs, take the (problem, K chunk) units of ONE row tile; the row-tile-per-XCD deal
  //  would leave every workgroup of a one-row-tile launch on XCD 0)
  const bool few = L.m_tiles < 8 && !(GH_DBG_BITS(L) & 32);
  const int m_tile = few ? bid / n_inner : xcd + 8 * (slot / n_inner);
  const int inner = few ? bid % n_inner : slot % n_inner;
  if (m_tile >= L.m_tiles) return;
  const int prob = inner % L.nprob;
  const int ks = inner / L.nprob;
  const bool split = L.ksplit > 1;
  const GH_KARG Problem& P = L.p[prob];
  const int M = P.M, N = P.N;
  const int m0 = m_tile * BM;
  if (m0 >= M) return;

  // measurement switches (tool build only, common.h): static wave priority per workgroup class, so that co-resident
  // workgroups drift out of phase
  const int dbg_bits = GH_DBG_BITS(L);
  if (dbg_bits & 4) {
    const int cls = ((bid >> 3) >> 5) % 3;
    if (cls == 0) __builtin_amdgcn_s_setprio(2); else if (cls == 1) __builtin_amdgcn_s_setprio(1);
  } else if (dbg_bits & 8) {
    const int cls = (bid >> 3) % 3;
    if (cls == 0) __builtin_amdgcn_s_setprio(2); else if (cls == 1) __builtin_amdgcn_s_setprio(1);
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int wrow = wm * 16 * MI, wcol = wn * 16 * NI;
  const int l15 = lane & 15, q = lane >> 4;

  const int nseg = P.nseg;
  const float* A0 = P.seg[0].A; const float* B0 = P.seg[0].B;
  const int lda0 = P.seg[0].lda, ldb0 = P.seg[0].ldb, K0 = P.seg[0].K;
  const float* A1 = nseg > 1 ? P.seg[1].A : A0; const float* B1 = nseg > 1 ? P.seg[1].B : B0;
  const int lda1 = nseg > 1 ? P.seg[1].lda : lda0, ldb1 = nseg > 1 ? P.seg[1].ldb : ldb0;
  const int K1 = nseg > 1 ? P.seg[1].K : 0;
  // K tiles are numbered over the concatenation of the segments; a K split (few-row GEMMs) gives this workgroup the tile
  // range [tbeg, tend) of that sequence -- chunks of L.kchunk / 16 tiles
  constexpr int kbeg = 0;
  const int kend = K0;
  const int nt0 = (K0 + BK - 1) / BK;
  // row tiles entirely at or beyond seg0_rows have an all-zero segment-0 operand (the aggregation of padding nodes in
  // the node-compact layout): start at the first tile of segment 1
  const int toff = (nseg > 1 && P.seg0_rows > 0 && m0 >= P.seg0_rows && !split) ? nt0 : 0;
  const int T_all = nt0 + (nseg > 1 ? (K1 + BK - 1) / BK : 0) - toff;
  int tbeg = 0, tend = (dbg_bits & 2) ? 1 : T_all;
  if (split) {
    const int ct = L.kchunk / BK;
    tbeg = ks * ct;
    tend = min(T_all, tbeg + ct);
    if (tbeg >= tend) return;
  }
  const int T = tend - tbeg;

  // ---- DMA slots of this wave: lane L of instruction i fills linear chunk 64 i + L = (row 16 i + L/4, slot L%4)
  const int drow = lane >> 2;
  const int dkq = (lane & 3) ^ nt_swz((lane >> 4) & 3);          // logical k-quad stored at this lane's slot
  unsigned a_vo0[SA], a_vo1[SA], b_vo0[SB], b_vo1[SB];
#pragma unroll
  for (int j = 0; j < SA; ++j) {
    const int ia = wave + NW * j;
    const int gm = m0 + 16 * ia + drow;
    const bool ok = (NW * (j + 1) <= NAI || ia < NAI) && gm < M;
    const int gmc = min(gm, M - 1);
    int s0 = gmc, s1 = gmc;
    if (P.seg[0].gatherA) s0 = P.seg[0].gatherA[gmc];              // embedding row id, read once per row
    if (nseg > 1 && P.seg[1].gatherA) s1 = P.seg[1].gatherA[gmc];
    a_vo0[j] = ok ? (unsigned)s0 * (unsigned)lda0 * (unsigned)ESZ + (unsigned)dkq * 16u : OOB;
    a_vo1[j] = ok ? (unsigned)s1 * (unsigned)lda1 * (unsigned)ESZ + (unsigned)dkq * 16u : OOB;
  }
#pragma unroll
  for (int j = 0; j < SB; ++j) {
    const int ib = wave + NW * j;
    if constexpr (X3P) {      // linear chunk c = (row c / 7, slot c % 7); slot 6 is the pitch padding
      const int c = ib * 64 + lane;
      const int n = c / 7, sl = c - 7 * n;
      const bool ok = (NW * (j + 1) <= NBI || ib < NBI) && n < N && n < BN && sl < 6;
      b_vo0[j] = ok ? (unsigned)n * (unsigned)ldb0 * 4u + (unsigned)sl * 16u : OOB;
      b_vo1[j] = ok ? (unsigned)n * (unsigned)ldb1 * 4u + (unsigned)sl * 16u : OOB;
    } else {
      const int n = 16 * ib + drow;
      const bool ok = (NW * (j + 1) <= NBI || ib < NBI) && n < N;
      b_vo0[j] = ok ? (unsigned)n * (unsigned)ldb0 * (unsigned)ESZ + (unsigned)dkq * 16u : OOB;
      b_vo1[j] = ok ? (unsigned)n * (unsigned)ldb1 * (unsigned)ESZ + (unsigned)dkq * 16u : OOB;
    }
  }

  const int drop_mode = P.drop_mode;
  const unsigned drop_seed = P.drop_seed, drop_thresh = P.drop_thresh;
  const float drop_scale = P.drop_scale;
  const int drop_ld = P.drop_ld;

  // one K tile: 16 k of BM rows of A and BN rows of B.  Invalid rows and k-quads at or beyond the segment's K carry
  // the out-of-range marker, for which the buffer range check returns 0 -> zeros land in LDS.
  auto dma_tile = [&](int t, int st) __attribute__((always_inline)) {
    const int tt = t + toff + tbeg;
    const bool s1 = tt >= nt0;
    const float* Ab = s1 ? A1 : A0;
    const float* Bb = s1 ? B1 : B0;
    const int k0 = s1 ? (tt - nt0) * BK : kbeg + tt * BK;
    const int klim = s1 ? K1 : kend;
    const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 0x7fffffff, 0x00020000);
    const rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, 0x7fffffff, 0x00020000);
    const bool kok = KQ * dkq < klim - k0;
    unsigned char* sb = smem + st * STAGE;
#pragma unroll
    for (int j = 0; j < SA; ++j) {
      const int ia = wave + NW * j;
      if (NW * (j + 1) <= NAI || ia < NAI)      // (as a streaming load -- aux = nt -- the A panel runs 2 % slower: measured)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(sb + ia * 1024), 16,
                                                 kok ? (s1 ? a_vo1[j] : a_vo0[j]) : OOB, k0 * ESZ, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < SB; ++j) {
      const int ib = wave + NW * j;
      if (NW * (j + 1) <= NBI || ib < NBI) {
        if constexpr (X3P)      // 6 bytes of split image per k; k beyond the segment's K meets zero A fragments (kok above)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(sb + BM * 64 + ib * 1024), 16,
                                                   s1 ? b_vo1[j] : b_vo0[j], k0 * 6, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(sb + BM * 64 + ib * 1024), 16,
                                                   kok ? (s1 ? b_vo1[j] : b_vo0[j]) : OOB, k0 * ESZ, 0, 0);
      }
    }
  };

  // ---- fragments: lane (l15, q) reads chunk (row l15, k-quad q) of a 16-row tile
  const int fsl = q ^ nt_swz((l15 >> 2) & 3);
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
  // swapped operands: acc[mi][ni][r] = C[row = wrow + mi*16 + l15][col = wcol + ni*16 + 4*q + r]
  auto mma_n = [&](const f32x4* a, const f32x4* b, int ni0, auto CNT) __attribute__((always_inline)) {
    constexpr int cnt = decltype(CNT)::value;
    if constexpr (MODE == 2) {
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int i = 0; i < cnt; ++i)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[mi][ni0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[i]), __builtin_bit_cast(bf16x8, a[mi]),
                                                                    acc[mi][ni0 + i], 0, 0, 0);
      return;
    }
    if constexpr (MODE == 3) {
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      // pieces of a fragment's four k values as packed bf16 pairs: {hi, mid, lo} x {k0k1, k2k3}
      auto split = [](const f32x4 v, unsigned* hi, unsigned* mid, unsigned* lo) __attribute__((always_inline)) {
        hi[0] = nt_pack_bf16(v[0], v[1]); hi[1] = nt_pack_bf16(v[2], v[3]);
        const float r0 = v[0] - __builtin_bit_cast(float, hi[0] << 16), r1 = v[1] - __builtin_bit_cast(float, hi[0] & 0xffff0000u);
        const float r2 = v[2] - __builtin_bit_cast(float, hi[1] << 16), r3 = v[3] - __builtin_bit_cast(float, hi[1] & 0xffff0000u);
        mid[0] = nt_pack_bf16(r0, r1); mid[1] = nt_pack_bf16(r2, r3);
        const float q0 = r0 - __builtin_bit_cast(float, mid[0] << 16), q1 = r1 - __builtin_bit_cast(float, mid[0] & 0xffff0000u);
        const float q2 = r2 - __builtin_bit_cast(float, mid[1] << 16), q3 = r3 - __builtin_bit_cast(float, mid[1] & 0xffff0000u);
        lo[0] = nt_pack_bf16(q0, q1); lo[1] = nt_pack_bf16(q2, q3);
      };
      auto cat = [](const unsigned* x, const unsigned* y) __attribute__((always_inline)) {
        const u32x4 u = {x[0], x[1], y[0], y[1]};
        return __builtin_bit_cast(bf16x8, u);
      };
      bf16x8 a_hm[MI], a_lh[MI], a_mh[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        unsigned h[2], m[2], l[2];
        split(a[mi], h, m, l);
        a_hm[mi] = cat(h, m); a_lh[mi] = cat(l, h); a_mh[mi] = cat(m, h);
      }
#pragma unroll
      for (int i = 0; i < cnt; ++i) {
        unsigned h[2], m[2], l[2];
        split(b[i], h, m, l);
        const bf16x8 b_hh = cat(h, h), b_hm = cat(h, m), b_ml = cat(m, l);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          acc[mi][ni0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_hh, a_hm[mi], acc[mi][ni0 + i], 0, 0, 0);   // ah.bh + am.bh
          acc[mi][ni0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_hm, a_lh[mi], acc[mi][ni0 + i], 0, 0, 0);   // al.bh + ah.bm
          acc[mi][ni0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_ml, a_mh[mi], acc[mi][ni0 + i], 0, 0, 0);   // am.bm + ah.bl
        }
      }
      return;
    }
    if constexpr (BF) {
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      s16x4 ab[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const uint2 u = make_uint2(nt_pack_bf16(a[mi][0], a[mi][1]), nt_pack_bf16(a[mi][2], a[mi][3]));
        ab[mi] = __builtin_bit_cast(s16x4, u);
      }
#pragma unroll
      for (int i = 0; i < cnt; ++i) {
        const uint2 u = make_uint2(nt_pack_bf16(b[i][0], b[i][1]), nt_pack_bf16(b[i][2], b[i][3]));
        const s16x4 bb = __builtin_bit_cast(s16x4, u);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[mi][ni0 + i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bb, ab[mi], acc[mi][ni0 + i], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < cnt; ++i)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[mi][ni0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i][s], a[mi][s], acc[mi][ni0 + i], 0, 0, 0);
  };
  // column tiles of this wave that hold at least one real column (N = 300: the last wave column has 9 of 10); the B
  // rows beyond N are zero in LDS, so skipping their MFMAs changes nothing but the time.  Wave-uniform.
#ifdef GH_NT_NOSKIP
  const int nv = NI;
#else
  const int nv = min(NI, max(0, (N - wcol + 15) >> 4));
#endif
  const int nvX = min(NH, nv), nvY = nv - nvX;
  auto mma = [&](const f32x4* a, const f32x4* b, int ni0, auto CNT) __attribute__((always_inline)) { mma_n(a, b, ni0, CNT); };
  // stateless input dropout (wrapper.py:189-190) on the A fragments of segment 0: element (row, k) of [rows][drop_ld]
  auto drop_a = [&](int t, f32x4* a) __attribute__((always_inline)) {
    const int tt = t + toff + tbeg;
    if (tt < nt0) {
      const int k = kbeg + tt * BK + KQ * q;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const unsigned idx = (unsigned)(m0 + wrow + mi * 16 + l15) * (unsigned)drop_ld + (unsigned)k;
        if constexpr (MODE == 2) {       // 8 bf16 per lane: the mask is exact in any precision (keep * 1/(1-p))
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          u32x4 w = __builtin_bit_cast(u32x4, a[mi]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float lo = __builtin_bit_cast(float, w[e] << 16), hi = __builtin_bit_cast(float, w[e] & 0xffff0000u);
            lo = drop_hash(drop_seed, idx + 2 * e) >= drop_thresh ? lo * drop_scale : 0.f;
            hi = drop_hash(drop_seed, idx + 2 * e + 1) >= drop_thresh ? hi * drop_scale : 0.f;
            w[e] = nt_pack_bf16(lo, hi);
          }
          a[mi] = __builtin_bit_cast(f32x4, w);
        } else {
          const float4 v = drop4(make_float4(a[mi][0], a[mi][1], a[mi][2], a[mi][3]), drop_seed, idx, drop_thresh, drop_scale);
          a[mi] = f32x4{v.x, v.y, v.z, v.w};
        }
      }
    }
  };

  // ---- main loop: 2 LDS stages, one barrier per K tile.  Instantiated for the common tile counts so that the MFMA
  //      stream is branch-free: every column tile valid / the last one of the Y batch all padding (N = 300).
  auto run = [&](auto CX, auto CY) __attribute__((always_inline)) {
    if constexpr (X3P) {
      // pre-split weights: the A fragments (8 floats per lane and K tile) are split in registers, the B pieces come out of LDS
      // as 8-byte reads; per 16 x 16 x 16 product three v_mfma_f32_16x16x32_bf16, each carrying two of the six terms.
      // The B reads are inline assembly with hand-placed lgkmcnt waits: the compiler puts s_waitcnt vmcnt(0) in front of
      // merged ds_read2_b64 reads that follow an LDS-DMA instruction (it cannot see that they touch the other stage), which
      // would serialise every K tile behind the prefetch of the next one.
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      constexpr int NX = decltype(CX)::value, NY = decltype(CY)::value;
      static_assert(NX <= 5 && NY <= 5, "wait_b below names five column tiles");
      const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
      const unsigned b3_fo = lds0 + (unsigned)(BM * 64) + (unsigned)(wcol + l15) * 112u + (unsigned)q * 24u;
      auto split = [](const f32x4 v, unsigned* hi, unsigned* mid, unsigned* lo) __attribute__((always_inline)) {
        hi[0] = nt_pack_bf16(v[0], v[1]); hi[1] = nt_pack_bf16(v[2], v[3]);
        const float r0 = v[0] - __builtin_bit_cast(float, hi[0] << 16), r1 = v[1] - __builtin_bit_cast(float, hi[0] & 0xffff0000u);
        const float r2 = v[2] - __builtin_bit_cast(float, hi[1] << 16), r3 = v[3] - __builtin_bit_cast(float, hi[1] & 0xffff0000u);
        mid[0] = nt_pack_bf16(r0, r1); mid[1] = nt_pack_bf16(r2, r3);
        const float q0 = r0 - __builtin_bit_cast(float, mid[0] << 16), q1 = r1 - __builtin_bit_cast(float, mid[0] & 0xffff0000u);
        const float q2 = r2 - __builtin_bit_cast(float, mid[1] << 16), q3 = r3 - __builtin_bit_cast(float, mid[1] & 0xffff0000u);
        lo[0] = nt_pack_bf16(q0, q1); lo[1] = nt_pack_bf16(q2, q3);
      };
      auto cat = [](const u32x2 x, const u32x2 y) __attribute__((always_inline)) {
        const u32x4 u = {x[0], x[1], y[0], y[1]};
        return __builtin_bit_cast(bf16x8, u);
      };
      struct Pieces { u32x2 h, m, l; };
      Pieces bx[5], by[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) { bx[i].h = bx[i].m = bx[i].l = u32x2{0u, 0u}; by[i] = bx[i]; }
      // pieces of `cnt` column tiles starting at tile ni0 of stage st (at most 15 reads in flight: lgkmcnt is a 4-bit counter)
      auto issue_b = [&](int st, Pieces* b, auto NI0, auto CNT) __attribute__((always_inline)) {
        constexpr int ni0 = decltype(NI0)::value, cnt = decltype(CNT)::value;
        const unsigned addr = b3_fo + (unsigned)(st * STAGE);
#pragma unroll
        for (int i = 0; i < cnt; ++i) {
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b[i].h) : "v"(addr), "n"((ni0 + i) * (16 * 112)));
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b[i].m) : "v"(addr), "n"((ni0 + i) * (16 * 112) + 8));
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b[i].l) : "v"(addr), "n"((ni0 + i) * (16 * 112) + 16));
        }
      };
      // every consumer of the pieces depends on this statement, so none can be scheduled above the wait
      auto wait_b = [](Pieces* b) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(b[0].h), "+v"(b[0].m), "+v"(b[0].l), "+v"(b[1].h), "+v"(b[1].m), "+v"(b[1].l), "+v"(b[2].h), "+v"(b[2].m),
                       "+v"(b[2].l), "+v"(b[3].h), "+v"(b[3].m), "+v"(b[3].l), "+v"(b[4].h), "+v"(b[4].m), "+v"(b[4].l)
                     :: "memory");
      };
      struct APieces { bf16x8 hm, lh, mh; };
      APieces apC[MI], apP[MI];
      auto mma_b = [&](const APieces* ap, const Pieces* b, auto NI0, auto CNT) __attribute__((always_inline)) {
        constexpr int ni0 = decltype(NI0)::value, cnt = decltype(CNT)::value;
#pragma unroll
        for (int i = 0; i < cnt; ++i) {
          const bf16x8 b_hh = cat(b[i].h, b[i].h), b_hm = cat(b[i].h, b[i].m), b_ml = cat(b[i].m, b[i].l);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[mi][ni0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_hh, ap[mi].hm, acc[mi][ni0 + i], 0, 0, 0);   // ah.bh + am.bh
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[mi][ni0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_hm, ap[mi].lh, acc[mi][ni0 + i], 0, 0, 0);   // al.bh + ah.bm
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[mi][ni0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_ml, ap[mi].mh, acc[mi][ni0 + i], 0, 0, 0);   // am.bm + ah.bl
        }
      };
      constexpr std::integral_constant<int, 0> I0{};
      constexpr std::integral_constant<int, NH> IH{};
      dma_tile(0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int t = 0; t < T; ++t) {
        const int st = t & 1;
        if (t + 1 < T) dma_tile(t + 1, st ^ 1);
        read_a(st, aC);
        issue_b(st, bx, I0, CX);
        __builtin_amdgcn_sched_barrier(0);
        if (t > 0) mma_b(apP, by, IH, CY);          // second half of tile t-1 covers the latency of the reads above
        if (drop_mode == 1) drop_a(t, aC);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          unsigned h[2], m[2], l[2];
          split(aC[mi], h, m, l);
          const u32x2 H = {h[0], h[1]}, Mm = {m[0], m[1]}, Lo = {l[0], l[1]};
          apC[mi].hm = cat(H, Mm); apC[mi].lh = cat(Lo, H); apC[mi].mh = cat(Mm, H);
        }
        __builtin_amdgcn_sched_barrier(0);
        wait_b(bx);
        issue_b(st, by, IH, CY);
        __builtin_amdgcn_sched_barrier(0);
        mma_b(apC, bx, I0, CX);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) apP[mi] = apC[mi];
        wait_b(by);                                  // this wave's reads of stage st are done before the barrier frees it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      mma_b(apP, by, IH, CY);
      return;
    }
    if constexpr (NST >= 3) {
      // NST stages, DMA prefetch distance NST - 1.  vmcnt counts DMA instructions: VM per K tile and wave, so "at most n
      // tiles still in flight" is vmcnt(n * VM); the barriers are bare (a __syncthreads() would drain the newest tiles too).
      static_assert(SA * NW == NAI && SB * NW == NBI && (SA + SB == 6 || SA + SB == 4 || SA + SB == 3) && NST <= 4, "vmcnt immediates below");
      constexpr int VM = SA + SB;
      auto wait_inflight = [&](int n) __attribute__((always_inline)) {      // n = tiles allowed to stay in flight (0..2)
        if (n >= 2) {
          if constexpr (VM == 6) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
          else if constexpr (VM == 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        } else if (n == 1) {
          if constexpr (VM == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
          else if constexpr (VM == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      };
#pragma unroll
      for (int i = 0; i < NST - 1; ++i)
        if (i < T) dma_tile(i, i);
      wait_inflight(min(T, NST - 1) - 1);            // tile 0 landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int st = 0;
      for (int t = 0; t < T; ++t) {
        const int stn = st == 0 ? NST - 1 : st - 1;      // the stage tile t-1 was read from, free since the last barrier
        if (t + NST - 1 < T) dma_tile(t + NST - 1, stn);
        read_a(st, aC);
        read_b(st, bX, 0, NH);
        __builtin_amdgcn_sched_barrier(0);
        if (t > 0) mma(aP, bY, NH, CY);
        __builtin_amdgcn_sched_barrier(0);
        read_b(st, bY, NH, NI - NH);
        if (drop_mode == 1) drop_a(t, aC);
        __builtin_amdgcn_sched_barrier(0);
        mma(aC, bX, 0, CX);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) aP[mi] = aC[mi];
        // tile t+1 landed (newer tiles may stay in flight), this wave's LDS reads of tile t are done
        wait_inflight(min(T - 1, t + NST - 1) - (t + 1));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        st = st == NST - 1 ? 0 : st + 1;
      }
      mma(aP, bY, NH, CY);
      __syncthreads();
      return;
    }
    dma_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      const int st = t & 1;
      if (t + 1 < T) dma_tile(t + 1, st ^ 1);
      read_a(st, aC);
      read_b(st, bX, 0, NH);
      __builtin_amdgcn_sched_barrier(0);
      if (t > 0) mma(aP, bY, NH, CY);              // second half of tile t-1 covers the latency of the reads above
      __builtin_amdgcn_sched_barrier(0);
      read_b(st, bY, NH, NI - NH);
      if (drop_mode == 1) drop_a(t, aC);
      __builtin_amdgcn_sched_barrier(0);
      mma(aC, bX, 0, CX);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) aP[mi] = aC[mi];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    mma(aP, bY, NH, CY);
  };
  // (any other count -- narrow problems, odd column blocks -- computes every tile: the B rows beyond N are zeros in LDS)
  if constexpr (PP) {
#include "gemm_nt_pp.hip.h"
  } else if constexpr (PP32) {
#include "gemm_nt_pp32.hip.h"
  } else {
    if (nvX == NH && nvY == NI - NH - 1) run(std::integral_constant<int, NH>{}, std::integral_constant<int, NI - NH - 1>{});
    else run(std::integral_constant<int, NH>{}, std::integral_constant<int, NI - NH>{});
  }

  // -------------------------------------------------------------------- epilogue on whole rows
  // Two passes (mi = 0, 1), each: accumulators (+bias) -> LDS, then a LINEAR pass over the 16*WM staged rows: item i
  // is float4 (row i / C4, column chunk i % C4) with C4 = EP_PITCH/4 a compile-time divisor, so a wave instruction
  // touches one contiguous run of a row in every epilogue stream.  Row reductions (attention head scores, the GSL
  // scorer's projection) read the finished rows back from LDS, one wave per row, in a fixed order (deterministic).
  if (dbg_bits & 1) {      // (tool build: K loop only.  Every accumulator stays live -- a test of two of them lets the compiler drop the other MFMAs)
    float sum = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) sum += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
    if (sum == 12345.678f) P.C[0] = 0.f;
    return;
  }
  const int epi = P.epi;
  const int ldc = P.ldc;
  float* const C = P.C + (split ? (size_t)ks * (size_t)P.split_stride : (size_t)0);
  const float* bias = P.bias;
  float* out1 = P.out1;
  const float* in0 = P.in0;
  const float* in1 = P.in1;
  const int accumulate = P.accumulate;
  const int heads = P.heads;
  const bool scorer = (epi == EPI_TANH_H) && (P.w2 != nullptr);
  const int pad_rows = (MODE == 2 || (dbg_bits & 64)) ? 0 : P.seg0_rows;                    // first padding row of the node-compact layout (0: none)
  const bool out_dead = scorer && P.ldu == 1;                            // EPI_TANH_H: padding rows' outputs feed the scorer only
  const bool rowred = (epi == EPI_ATT) || scorer;
  float* ep = reinterpret_cast<float*>(smem);
  float* const ep_bias_ptr = ep + (PP ? 64 : 16 * WM) * EP_PITCH + NW * 64 * 4;      // (PP: behind the 64 staged rows of gemm_nt_pp_epi.hip.h)
  const int N4 = N >> 2;
  constexpr int C4 = EP_PITCH / 4;
  constexpr int ITEMS = 16 * WM * C4;
  constexpr int NIT = (ITEMS + NTHR - 1) / NTHR;
  const float* bias2 = P.bias2;
  // bias (sum) staged in LDS behind the row-reduction partials: the linear pass reads it next to the staged row
  float* bsum = ep_bias_ptr;
  for (int c = tid; c < BN; c += NTHR) {
    float bv = 0.f;
    if (c < N) { if (bias) bv = bias[c]; if (bias2) bv += bias2[c]; }
    bsum[c] = bv;
  }
  // epilogue streams are fp32, or (MODE 2) bf16 per Problem::io
  const int io = (MODE == 2) ? P.io : 0;
  float* const c32 = (MODE == 2) ? P.c32 : nullptr;
  auto ld4 = [&](const float* p, size_t o, bool bf) __attribute__((always_inline)) {
    if constexpr (MODE == 2) {
      if (bf) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + o);
        return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                           __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
      }
    }
    return *reinterpret_cast<const float4*>(p + o);
  };
  auto st4 = [&](float* p, size_t o, const float4 v, bool bf) __attribute__((always_inline)) {
    if constexpr (MODE == 2) {
      if (bf) {
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p) + o) = make_uint2(nt_pack_bf16(v.x, v.y), nt_pack_bf16(v.z, v.w));
        return;
      }
    }
    // streaming (nontemporal) stores: an epilogue output is ~75 MB that the next launch reads from the start; keeping its
    // tail in L2 only evicts the weight panels the K loops are re-reading (A/B on the bench step: +0.4 %; nontemporal LOADS
    // of the epilogue inputs: no change)
    if (!(dbg_bits & 16)) { __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p + o)); return; }
    *reinterpret_cast<float4*>(p + o) = v;
  };
  auto row_reduce = [&](auto MIT) __attribute__((always_inline)) {
    constexpr int mi = decltype(MIT)::value;
    if (rowred) {
      // e[row][c] = sum_k y[row][k] w2[c][k] for the 16*WM finished rows in LDS: a [16 x N] x [N x <=8] product per row
      // tile, on MFMA.  Wave (b, kp) takes row tile b and every KSPL-th K tile; partials meet in LDS (fixed order).
      __syncthreads();
      constexpr int KSPL = NW / WM;
      const int nred = scorer ? 1 : heads;
      const int b = wave % WM, kp = wave / WM;
      const float* arow = ep + (b * 16 + l15) * EP_PITCH + 4 * q;
      // (EPI_ATT on a column block of a wider row -- Batch::add, att_blocks: w2 is [heads][ldu], this block starts at its column)
      const float* wrow = P.w2 + (size_t)min(l15, nred - 1) * (epi == EPI_ATT ? P.ldu : N) + 4 * q;
      f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int k0 = kp * 16; k0 < N; k0 += 16 * KSPL) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + k0);          // columns >= N of the staged tile are exact zeros
        f32x4 w4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (l15 < nred && k0 + 4 * q < N) w4 = *reinterpret_cast<const f32x4*>(wrow + k0);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) d = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[s2], a4[s2], d, 0, 0, 0);
      }
      // d[r] = partial e[row l15][c = 4q + r]
      float* red = ep + 16 * WM * EP_PITCH;                                   // [KSPL][WM][64] float4, behind the staged rows
      *reinterpret_cast<f32x4*>(red + ((kp * WM + b) * 64 + lane) * 4) = d;
      __syncthreads();
      if (kp == 0) {
        f32x4 sum = d;
#pragma unroll
        for (int k2 = 1; k2 < KSPL; ++k2) sum += *reinterpret_cast<const f32x4*>(red + ((k2 * WM + b) * 64 + lane) * 4);
        const int row = m0 + b * 16 * MI + mi * 16 + l15;
        if (row < M) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * q + r < nred) {
              if (P.e_atomic == 1) atomicAdd(&P.e[(size_t)row * nred + 4 * q + r], sum[r]);
              else P.e[(size_t)row * nred + 4 * q + r] = sum[r];
            }
        }
      }
    }
  };
  auto epilogue_pass = [&](auto MIT) __attribute__((always_inline)) {
    constexpr int mi = decltype(MIT)::value;
    if (mi > 0) __syncthreads();                  // previous pass consumed (the loop's last barrier covers pass 0)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      *reinterpret_cast<f32x4*>(ep + (wm * 16 + l15) * EP_PITCH + wcol + ni * 16 + 4 * q) = acc[mi][ni];
    __syncthreads();
    // the pass holds, for every wave row b < WM, tile rows b*16*MI + mi*16 + [0,16).  Items are processed CH at a
    // time: every load of the chunk (LDS + the epilogue's input streams) is issued before the first store -- the
    // pointers may alias as far as the compiler knows, so a plain loop would serialise 11 memory round trips per pass.
    auto pass = [&](auto EPI) __attribute__((always_inline)) {
      constexpr int E = decltype(EPI)::value;
#ifndef GH_EPI_CH
#define GH_EPI_CH 4
#endif
      constexpr int CH = GH_EPI_CH;
#pragma unroll
      for (int it0 = 0; it0 < NIT; it0 += CH) {
        float4 xa[CH], xb[CH], xc[CH], xd[E == EPI_GATE_PRE ? CH : 1];
        auto where = [&](int j, int& row, int& col, float*& sp) __attribute__((always_inline)) {
          const int i = tid + (it0 + j) * NTHR;
          const int rr = i / C4, c4 = i - rr * C4;
          row = m0 + (rr >> 4) * 16 * MI + mi * 16 + (rr & 15);
          col = 4 * c4;
          sp = ep + rr * EP_PITCH + col;
          return (it0 + j < NIT) && (ITEMS % NTHR == 0 || i < ITEMS) && c4 < N4 && row < M;
        };
        if (E != EPI_SIGMOID_Z) {
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            int row, col; float* sp;
            if (where(j, row, col, sp)) {
              const size_t o = (size_t)row * ldc + col;
              if (E == EPI_STORE) { if (accumulate) xa[j] = ld4(C, o, io & 1); }
              else if (E == EPI_SIGMOID_R) xa[j] = ld4(in0, o, io & 4);
              else if (E == EPI_TANH_H) { xa[j] = ld4(in0, o, io & 4); xb[j] = ld4(in1, o, io & 8); }
              else if (E == EPI_BWD_DRX) {
                xa[j] = ld4(in0, o, io & 4);
                xb[j] = ld4(in1, o, io & 8);
                xc[j] = ld4(out1, o, io & 2);
              } else if (E == EPI_GATE_PRE) {
                xa[j] = ld4(in0, o, io & 4);
                xb[j] = ld4(in1, o, io & 8);
                xc[j] = ld4(P.in2, o, io & 16);
                if (P.gin) xd[j] = ld4(P.gin, o, false);      // (the addend of g is always fp32)
              } else if (E == EPI_ATT)
                xa[j] = *reinterpret_cast<const float4*>(P.u + (size_t)(P.rowg ? P.rowg[row] : row / P.R) * P.ldu + col);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          int row, col; float* sp;
          if (!where(j, row, col, sp)) continue;
          const float4 v4 = *reinterpret_cast<const float4*>(sp);
          const float4 b4 = *reinterpret_cast<const float4*>(bsum + col);
          float4 w = make_float4(v4.x + b4.x, v4.y + b4.y, v4.z + b4.z, v4.w + b4.w);
          const size_t o = (size_t)row * ldc + col;
          if (E == EPI_STORE) {
            if (drop_mode == 3)
              w = drop4(w, drop_seed, (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col), drop_thresh, drop_scale);
            if (accumulate) { w.x += xa[j].x; w.y += xa[j].y; w.z += xa[j].z; w.w += xa[j].w; }
            st4(C, o, w, io & 1);
          } else if (E == EPI_SIGMOID_Z) {
            st4(C, o, make_float4(sigmoidf_(w.x), sigmoidf_(w.y), sigmoidf_(w.z), sigmoidf_(w.w)), io & 1);
          } else if (E == EPI_SIGMOID_R) {
            const float4 x = xa[j];
            const float4 r4 = make_float4(sigmoidf_(w.x), sigmoidf_(w.y), sigmoidf_(w.z), sigmoidf_(w.w));
            // padding rows of the node-compact layout (row >= seg0_rows): r is read by the backward only, and the backward
            // never touches a padding row -- the store is dead (a third of the first cell's rows in training mode)
            if (!(pad_rows > 0 && row >= pad_rows)) st4(C, o, r4, io & 1);
            st4(out1, o, make_float4(r4.x * x.x, r4.y * x.y, r4.z * x.z, r4.w * x.w), io & 2);
          } else if (E == EPI_TANH_H) {
            const float4 z = xa[j], x = xb[j];
            const float4 h = make_float4(tanhf_(w.x), tanhf_(w.y), tanhf_(w.z), tanhf_(w.w));
            float4 y = make_float4(h.x * z.x + x.x * (1.f - z.x), h.y * z.y + x.y * (1.f - z.y),
                                   h.z * z.z + x.z * (1.f - z.z), h.w * z.w + x.w * (1.f - z.w));
            // padding rows: h~ is backward-only (dead, as r above); the cell output of a padding row feeds the fused scorer
            // projection below and nothing else when the caller says so (ldu = 1: the composite forward, whose second cell
            // and attention run on the real rows only)
            const bool pad = pad_rows > 0 && row >= pad_rows;
            if (!pad) st4(C, o, h, io & 1);
            if (!(pad && out_dead)) st4(out1, o, y, io & 2);
            if (c32) *reinterpret_cast<float4*>(c32 + o) = y;
            if (scorer) {    // the word scorer sees dropout(out) (its own input dropout, wrapper.py:189-190)
              if (drop_mode == 2)
                y = drop4(y, drop_seed, (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col), drop_thresh, drop_scale);
              *reinterpret_cast<float4*>(sp) = y;
            }
          } else if (E == EPI_BWD_DRX) {
            const float4 x = xa[j], r4 = xb[j];
            float4 d = xc[j];
            st4(C, o, make_float4(w.x * x.x * r4.x * (1.f - r4.x), w.y * x.y * r4.y * (1.f - r4.y),
                                  w.z * x.z * r4.z * (1.f - r4.z), w.w * x.w * r4.w * (1.f - r4.w)), io & 1);
            d.x += w.x * r4.x; d.y += w.y * r4.y; d.z += w.z * r4.z; d.w += w.w * r4.w;
            st4(out1, o, d, io & 2);
          } else if (E == EPI_GATE_PRE) {
            if (drop_mode == 3)      // g is the gradient w.r.t. a dropped-out input (the producing cell's dX)
              w = drop4(w, drop_seed, (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col), drop_thresh, drop_scale);
            if (P.gin) { const float4 a4 = xd[j]; w.x += a4.x; w.y += a4.y; w.z += a4.z; w.w += a4.w; }
            const float4 Z = xa[j], Hh = xb[j], X = xc[j];
            float4 a, b, c;
#define GH_ONE(f)                                      \
            a.f = w.f * Z.f * (1.f - Hh.f * Hh.f);             \
            b.f = w.f * (Hh.f - X.f) * Z.f * (1.f - Z.f);      \
            c.f = w.f * (1.f - Z.f);
            GH_ONE(x) GH_ONE(y) GH_ONE(z) GH_ONE(w)
#undef GH_ONE
            if constexpr (MODE == 2) {      // bf16 storage: the gate head's scratch holds bf16 (io bits 1 / 2 / 32)
              st4(C, o, a, io & 1); st4(out1, o, b, io & 2); st4(P.out2, o, c, io & 32);
            } else {
              *reinterpret_cast<float4*>(C + o) = a;
              *reinterpret_cast<float4*>(out1 + o) = b;
              *reinterpret_cast<float4*>(P.out2 + o) = c;
            }
          } else if (E == EPI_ATT) {
            const float4 u4 = xa[j];
            const float4 t4 = make_float4(tanhf_(w.x + u4.x), tanhf_(w.y + u4.y), tanhf_(w.z + u4.z), tanhf_(w.w + u4.w));
            *reinterpret_cast<float4*>(C + o) = t4;
            *reinterpret_cast<float4*>(sp) = t4;
          }
        }
      }
    };
    // bf16 storage pipeline (MODE 2): the same linear pass on items of EIGHT columns -- one 16-byte access per bf16
    // stream and item (two float4 of the staged row) instead of the 8-byte accesses a float4 item gives: the epilogue of
    // the 128 x 256 tile was the larger half of its launch at ~2.5 TB/s.  Column blocks of the bf16
    // pipeline are multiples of 8 wide (h % 8 == 0).
    auto pass8 = [&](auto EPI) __attribute__((always_inline)) {
      constexpr int E = decltype(EPI)::value;
      constexpr int C8 = BN / 8;
      constexpr int ITEMS8 = 16 * WM * C8;
      constexpr int NIT8 = (ITEMS8 + NTHR - 1) / NTHR;
      constexpr int CH = 2;
      struct F8 { float4 a, b; };
      auto ld8 = [&](const float* p, size_t o, bool bf) __attribute__((always_inline)) {
        F8 r;
        if (bf) {
          const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p) + o);
          r.a = make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                            __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
          r.b = make_float4(__builtin_bit_cast(float, u.z << 16), __builtin_bit_cast(float, u.z & 0xffff0000u),
                            __builtin_bit_cast(float, u.w << 16), __builtin_bit_cast(float, u.w & 0xffff0000u));
        } else {
          r.a = *reinterpret_cast<const float4*>(p + o);
          r.b = *reinterpret_cast<const float4*>(p + o + 4);
        }
        return r;
      };
      auto st8 = [&](float* p, size_t o, const float4 a, const float4 b, bool bf) __attribute__((always_inline)) {
        if (bf) {
          *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p) + o) =
              make_uint4(nt_pack_bf16(a.x, a.y), nt_pack_bf16(a.z, a.w), nt_pack_bf16(b.x, b.y), nt_pack_bf16(b.z, b.w));
        } else {
          *reinterpret_cast<float4*>(p + o) = a;
          *reinterpret_cast<float4*>(p + o + 4) = b;
        }
      };
      auto sig4 = [](const float4 v) __attribute__((always_inline)) { return make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)); };
      auto tanh4 = [](const float4 v) __attribute__((always_inline)) { return make_float4(tanhf_(v.x), tanhf_(v.y), tanhf_(v.z), tanhf_(v.w)); };
      auto mul4 = [](const float4 a, const float4 b) __attribute__((always_inline)) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); };
      auto add4 = [](const float4 a, const float4 b) __attribute__((always_inline)) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
      auto mix4 = [](const float4 h, const float4 z, const float4 x) __attribute__((always_inline)) {      // h z + x (1 - z)
        return make_float4(h.x * z.x + x.x * (1.f - z.x), h.y * z.y + x.y * (1.f - z.y), h.z * z.z + x.z * (1.f - z.z), h.w * z.w + x.w * (1.f - z.w));
      };
      auto drx4 = [](const float4 w, const float4 x, const float4 r) __attribute__((always_inline)) {      // w x r (1 - r)
        return make_float4(w.x * x.x * r.x * (1.f - r.x), w.y * x.y * r.y * (1.f - r.y), w.z * x.z * r.z * (1.f - r.z), w.w * x.w * r.w * (1.f - r.w));
      };
#pragma unroll
      for (int it0 = 0; it0 < NIT8; it0 += CH) {
        F8 xa[CH], xb[CH], xc[CH], xd[E == EPI_GATE_PRE ? CH : 1];
        auto where = [&](int j, int& row, int& col, float*& sp) __attribute__((always_inline)) {
          const int i = tid + (it0 + j) * NTHR;
          const int rr = i / C8, c8 = i - rr * C8;
          row = m0 + (rr >> 4) * 16 * MI + mi * 16 + (rr & 15);
          col = 8 * c8;
          sp = ep + rr * EP_PITCH + col;
          return (it0 + j < NIT8) && (ITEMS8 % NTHR == 0 || i < ITEMS8) && col < N && row < M;
        };
        if (E != EPI_SIGMOID_Z) {
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            int row, col; float* sp;
            if (where(j, row, col, sp)) {
              const size_t o = (size_t)row * ldc + col;
              if (E == EPI_STORE) { if (accumulate) xa[j] = ld8(C, o, io & 1); }
              else if (E == EPI_SIGMOID_R) xa[j] = ld8(in0, o, io & 4);
              else if (E == EPI_TANH_H) { xa[j] = ld8(in0, o, io & 4); xb[j] = ld8(in1, o, io & 8); }
              else if (E == EPI_BWD_DRX) { xa[j] = ld8(in0, o, io & 4); xb[j] = ld8(in1, o, io & 8); xc[j] = ld8(out1, o, io & 2); }
              else if (E == EPI_GATE_PRE) {
                xa[j] = ld8(in0, o, io & 4); xb[j] = ld8(in1, o, io & 8); xc[j] = ld8(P.in2, o, io & 16);
                if (P.gin) xd[j] = ld8(P.gin, o, false);
              }
            }
          }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          int row, col; float* sp;
          if (!where(j, row, col, sp)) continue;
          float4 wa = add4(*reinterpret_cast<const float4*>(sp), *reinterpret_cast<const float4*>(bsum + col));
          float4 wb = add4(*reinterpret_cast<const float4*>(sp + 4), *reinterpret_cast<const float4*>(bsum + col + 4));
          const size_t o = (size_t)row * ldc + col;
          if (E == EPI_STORE) {
            if (drop_mode == 3) {
              const unsigned idx = (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col);
              wa = drop4(wa, drop_seed, idx, drop_thresh, drop_scale);
              wb = drop4(wb, drop_seed, idx + 4u, drop_thresh, drop_scale);
            }
            if (accumulate) { wa = add4(wa, xa[j].a); wb = add4(wb, xa[j].b); }
            st8(C, o, wa, wb, io & 1);
          } else if (E == EPI_SIGMOID_Z) {
            st8(C, o, sig4(wa), sig4(wb), io & 1);
          } else if (E == EPI_SIGMOID_R) {
            const float4 ra = sig4(wa), rb = sig4(wb);
            st8(C, o, ra, rb, io & 1);
            st8(out1, o, mul4(ra, xa[j].a), mul4(rb, xa[j].b), io & 2);
          } else if (E == EPI_TANH_H) {
            const float4 ha = tanh4(wa), hb = tanh4(wb);
            const float4 ya = mix4(ha, xa[j].a, xb[j].a), yb = mix4(hb, xa[j].b, xb[j].b);
            st8(C, o, ha, hb, io & 1);
            st8(out1, o, ya, yb, io & 2);
            if (c32) { *reinterpret_cast<float4*>(c32 + o) = ya; *reinterpret_cast<float4*>(c32 + o + 4) = yb; }
            if (scorer) {    // the word scorer sees dropout(out) in fp32 (its own input dropout, wrapper.py:189-190): staged for the row reduction
              float4 sa = ya, sb = yb;
              if (drop_mode == 2) {
                const unsigned idx = (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col);
                sa = drop4(sa, drop_seed, idx, drop_thresh, drop_scale);
                sb = drop4(sb, drop_seed, idx + 4u, drop_thresh, drop_scale);
              }
              *reinterpret_cast<float4*>(sp) = sa;
              *reinterpret_cast<float4*>(sp + 4) = sb;
            }
          } else if (E == EPI_BWD_DRX) {
            st8(C, o, drx4(wa, xa[j].a, xb[j].a), drx4(wb, xa[j].b, xb[j].b), io & 1);
            st8(out1, o, add4(xc[j].a, mul4(wa, xb[j].a)), add4(xc[j].b, mul4(wb, xb[j].b)), io & 2);
          } else if (E == EPI_GATE_PRE) {      // g = w (+ gin), masked when g is the gradient w.r.t. a dropped-out input: the gate head of the cell below
            if (drop_mode == 3) {
              const unsigned idx = (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col);
              wa = drop4(wa, drop_seed, idx, drop_thresh, drop_scale);
              wb = drop4(wb, drop_seed, idx + 4u, drop_thresh, drop_scale);
            }
            if (P.gin) { wa = add4(wa, xd[j].a); wb = add4(wb, xd[j].b); }
            auto head = [](const float4 g, const float4 Z, const float4 Hh, const float4 X, float4& a, float4& b, float4& c) __attribute__((always_inline)) {
#define GH_ONE8(f)                                    \
              a.f = g.f * Z.f * (1.f - Hh.f * Hh.f);          \
              b.f = g.f * (Hh.f - X.f) * Z.f * (1.f - Z.f);   \
              c.f = g.f * (1.f - Z.f);
              GH_ONE8(x) GH_ONE8(y) GH_ONE8(z) GH_ONE8(w)
#undef GH_ONE8
            };
            float4 a0, b0, c0, a1, b1, c1;
            head(wa, xa[j].a, xb[j].a, xc[j].a, a0, b0, c0);
            head(wb, xa[j].b, xb[j].b, xc[j].b, a1, b1, c1);
            st8(C, o, a0, a1, io & 1); st8(out1, o, b0, b1, io & 2); st8(P.out2, o, c0, c1, io & 32);
          }
        }
      }
    };
    constexpr bool WIDE8 = (MODE == 2) && (BN % 8 == 0);
    bool done8 = false;
    if constexpr (WIDE8) {
      if (epi != EPI_ATT) {      // (the fused scorer projection -- EPI_TANH_H with w2 -- stages its rows here and reduces them below)
        if (epi == EPI_STORE) pass8(std::integral_constant<int, EPI_STORE>{});
        else if (epi == EPI_GATE_PRE) pass8(std::integral_constant<int, EPI_GATE_PRE>{});
        else if (epi == EPI_SIGMOID_Z) pass8(std::integral_constant<int, EPI_SIGMOID_Z>{});
        else if (epi == EPI_SIGMOID_R) pass8(std::integral_constant<int, EPI_SIGMOID_R>{});
        else if (epi == EPI_TANH_H) pass8(std::integral_constant<int, EPI_TANH_H>{});
        else if (epi == EPI_BWD_DRX) pass8(std::integral_constant<int, EPI_BWD_DRX>{});
        if (!rowred) return;
        done8 = true;
      }
    }
    if (done8) { /* rows staged by pass8 */ }
    else if (epi == EPI_STORE) pass(std::integral_constant<int, EPI_STORE>{});
    else if (epi == EPI_SIGMOID_Z) pass(std::integral_constant<int, EPI_SIGMOID_Z>{});
    else if (epi == EPI_SIGMOID_R) pass(std::integral_constant<int, EPI_SIGMOID_R>{});
    else if (epi == EPI_TANH_H) pass(std::integral_constant<int, EPI_TANH_H>{});
    else if (epi == EPI_BWD_DRX) pass(std::integral_constant<int, EPI_BWD_DRX>{});
    else if (epi == EPI_GATE_PRE) pass(std::integral_constant<int, EPI_GATE_PRE>{});
    else if (epi == EPI_ATT) pass(std::integral_constant<int, EPI_ATT>{});
    row_reduce(MIT);
  };
  if constexpr ((MODE == 0 || MODE == 1) && MI == 2) {      // (the experimental fp32x3 modes keep the plain passes: build time)
    bool f32_done = false;
#include "gemm_nt_epi32.hip.h"
    if (f32_done) return;
  }
  if constexpr (PP) {
    bool pp_done = true;
#include "gemm_nt_pp_epi.hip.h"
    if (pp_done) return;
  }
  epilogue_pass(std::integral_constant<int, 0>{});
  epilogue_pass(std::integral_constant<int, 1>{});
  if constexpr (MI >= 4) {
    epilogue_pass(std::integral_constant<int, 2>{});
    epilogue_pass(std::integral_constant<int, 3>{});
  }
  if constexpr (MI == 8) {
    epilogue_pass(std::integral_constant<int, 4>{});
    epilogue_pass(std::integral_constant<int, 5>{});
    epilogue_pass(std::integral_constant<int, 6>{});
    epilogue_pass(std::integral_constant<int, 7>{});
  }
#endif
}

}  // namespace gh
