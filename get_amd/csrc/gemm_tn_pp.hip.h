// bf16-storage weight-gradient GEMM on the ping-pong loop of gemm_nt_pp.hip.h (round 6; prototype with its host check and the
// measurements: tools/tn256_proto.hip):
//     Cpart_p[ks][I][J] (fp32) = sum_{m in K chunk ks} A_p[m][I] . B_p[m][J]          A = upstream gradient rows, B = activation rows
// 256 x 256 x 64 tile, 8 waves (2 x 4; a wave's 128 x 64 outputs as four 64 x 32 quadrants), ONE workgroup per CU, the two wave rows
// one barrier interval apart: a K tile is four phases of {load section: transpose reads of one quadrant's new operands + the LDS-DMA
// of one half-tile of a later K tile; barrier; MFMA section: 16 x v_mfma_f32_16x16x32_bf16; barrier}, counted vmcnt waits, two
// 64 KB buffers.  What differs from the NT loop is the LDS image and the fragment read -- the operands are k-major in memory:
//   * a half-tile is 64 k-rows x 128 columns, 256 B per k-row = two whole cache lines of the operand row.  Half h of A = output rows
//     i0 + 128 h + [0, 128) (wave row wm owns 64 wm + [0, 64) of them), half h of B = output columns j0 + 128 h + [0, 128) (wave
//     column wn owns 32 wn + [0, 32)): the wave's outputs are not contiguous, the halves of the operand rows are;
//   * the MFMA wants 8 consecutive k per lane, the transpose of the image: two ds_read_b64_tr_b16 per fragment (semantics measured in
//     tools/tr_probe.hip, as gemm_tn.hip.h).  Every k-row of a 256-byte-pitch image starts on the same bank, so 32-byte units of a
//     k-row are XOR-swizzled with (r & 3) | ((r >> 3) & 1) << 2 on the SOURCE side of the DMA: the eight k-rows {0..3, 8..11} (+4, +16)
//     that one 32-lane half of a transpose read touches sit on eight different bank octets.
//   * rows >= kend of a chunk's last K tile lie beyond the buffer descriptor's num_records and read as zeros (the K offset rides in
//     the vector offset); stages beyond the last tile carry the out-of-range marker so that the vmcnt bookkeeping stays uniform.
// Work items (K chunk, problem, row tile, column tile) are dealt to the XCDs in RUNS -- XCD x takes items [x per, (x + 1) per) of the
// chunk-major list -- so that the workgroups sharing an L2 stream the same rows of the same operands.
// The bias gradient (column sums of A over the chunk) comes out of the A fragments the waves hold anyway: wave (wm, wn) sums tile
// mi = wn of each row half with one more MFMA per fragment against an all-ones operand (4 per K tile, in the tiles that carry one).
// Measured (prototype, K = 62 208 rows): 768 x 1536 and 1536 x 1536 outputs 1030 - 1100 TF with their partial-tile stores, against
// ~840 TF for the 128 x 320 kernel's cell launches inside the step.
// Host-side preconditions (Batch::flush): bf16 operands, 16-byte aligned rows, I and J multiples of 8, no row gather, partial tiles in
// the split-K workspace (plain stores), 31-bit byte offsets.
#pragma once
#include "gemm.hip.h"
#include <type_traits>

namespace gh {

__global__ void __launch_bounds__(512, 1) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_tn_pp_kernel(const Launch L_byval) {
  (void)L_byval;
#if defined(__HIP_DEVICE_COMPILE__)
  const GH_KARG Launch& L = *(const GH_KARG Launch*)__builtin_amdgcn_kernarg_segment_ptr();
  typedef __amdgpu_buffer_rsrc_t rsrc_t;
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr unsigned OOB = 0x80000000u;
  constexpr int BUF = 65536, BOFF = 32768, HALF = 16384;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, q = lane >> 4;

  const int per_prob = L.m_tiles * L.n_tiles;
  const int ntiles = per_prob * L.nprob;
  const int bid = blockIdx.x;
  const int w = (bid & 7) * L.per + (bid >> 3);
  if ((bid >> 3) >= L.per || w >= ntiles * L.ksplit) return;
  const int ks = w / ntiles, tile = w % ntiles;
  const int prob = tile / per_prob, rem = tile % per_prob;
  const int jt = rem % L.n_tiles;
  const GH_KARG Problem& P = L.p[prob];
  const int I = P.M, J = P.N;
  const int i0 = (rem / L.n_tiles) * 256, j0 = jt * 256;
  if (i0 >= I || j0 >= J) return;
  const int kbeg = ks * L.kchunk;
  const int kend = min(P.seg[0].K, kbeg + L.kchunk);
  const int T = kend > kbeg ? (kend - kbeg + 63) / 64 : 0;      // (an empty chunk still writes its zero tile: the reduction reads it)
  const int lda = P.seg[0].lda, ldb = P.seg[0].ldb;

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
      a_vo[h][j] = ci < I ? ((unsigned)(kbeg + r) * (unsigned)lda + (unsigned)ci) * 2u : OOB;
      b_vo[h][j] = cj < J ? ((unsigned)(kbeg + r) * (unsigned)ldb + (unsigned)cj) * 2u : OOB;
    }
  const int kend_c = max(kend, 0);
  const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)P.seg[0].A, 0, kend_c * lda * 2, 0x00020000);
  const rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)P.seg[0].B, 0, kend_c * ldb * 2, 0x00020000);
  const unsigned a_step = 64u * (unsigned)lda * 2u, b_step = 64u * (unsigned)ldb * 2u;
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
  unsigned a_ad[2][4], b_ad[2][2];      // [buffer][16-column tile of the wave's share of a half]
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
  uint2 aF[4][2][2], bF[2][2][2][2];      // A: [mi][k-step][read] of the current row half; B: [column half][ni][k-step][read]
  // [row half]: column sums of A (the bias gradient) for the wave's tile mi = wn, as one more MFMA per fragment against an all-ones B
  // operand -- every output column of the 16 x 16 result then holds sum_k A[k][i]
  f32x4 acs[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const bf16x8 ones8 = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
  float* const colsum = P.colsum;
  // (the two row halves' sums ride in two DIFFERENT column tiles of the row tile when the output has more than one: the tiles that
  //  carry a bias gradient then run 2 instead of 4 extra MFMAs per K tile, and a launch is as slow as its slowest workgroup)
  const bool do_cs0 = colsum != nullptr && jt == 0;
  const bool do_cs1 = colsum != nullptr && jt == (J > 256 ? 1 : 0);

#define GH_TP_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define GH_TP_RD_A(B_, H_)                                                            \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                  \
    GH_TP_TR(aF[mi][0][0], a_ad[B_][mi], (H_) * 16384 + 0);                           \
    GH_TP_TR(aF[mi][0][1], a_ad[B_][mi], (H_) * 16384 + 1024);                        \
    GH_TP_TR(aF[mi][1][0], a_ad[B_][mi], (H_) * 16384 + 8192);                        \
    GH_TP_TR(aF[mi][1][1], a_ad[B_][mi], (H_) * 16384 + 8192 + 1024);                 \
  }
#define GH_TP_RD_B(B_, H_)                                                            \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                  \
    GH_TP_TR(bF[H_][ni][0][0], b_ad[B_][ni], (H_) * 16384 + 0);                       \
    GH_TP_TR(bF[H_][ni][0][1], b_ad[B_][ni], (H_) * 16384 + 1024);                    \
    GH_TP_TR(bF[H_][ni][1][0], b_ad[B_][ni], (H_) * 16384 + 8192);                    \
    GH_TP_TR(bF[H_][ni][1][1], b_ad[B_][ni], (H_) * 16384 + 8192 + 1024);             \
  }
#define GH_TP_FRAG(F_) __builtin_bit_cast(bf16x8, make_uint4((F_)[0].x, (F_)[0].y, (F_)[1].x, (F_)[1].y))
#define GH_TP_QUAD(RH_, CH_)                                                          \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                    \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                  \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                \
        acc[(RH_) * 4 + mi][(CH_) * 2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16( \
            GH_TP_FRAG(bF[CH_][ni][kk]), GH_TP_FRAG(aF[mi][kk]), acc[(RH_) * 4 + mi][(CH_) * 2 + ni], 0, 0, 0);
  // the transpose reads are invisible to the compiler's counters: the wait lists every destination as an in/out operand, so that no use
  // of a register that has not landed yet can be scheduled above it (cdna_hip_programming.md, inline-asm form ii)
#define GH_TP_WAIT_A() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aF[0][0][0]), "+v"(aF[0][0][1]), "+v"(aF[0][1][0]), "+v"(aF[0][1][1]), \
    "+v"(aF[1][0][0]), "+v"(aF[1][0][1]), "+v"(aF[1][1][0]), "+v"(aF[1][1][1]), "+v"(aF[2][0][0]), "+v"(aF[2][0][1]), "+v"(aF[2][1][0]), "+v"(aF[2][1][1]), \
    "+v"(aF[3][0][0]), "+v"(aF[3][0][1]), "+v"(aF[3][1][0]), "+v"(aF[3][1][1]) :: "memory")
#define GH_TP_WAIT_B(H_) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bF[H_][0][0][0]), "+v"(bF[H_][0][0][1]), "+v"(bF[H_][0][1][0]), "+v"(bF[H_][0][1][1]), \
    "+v"(bF[H_][1][0][0]), "+v"(bF[H_][1][0][1]), "+v"(bF[H_][1][1][0]), "+v"(bF[H_][1][1][1]) :: "memory")
#define GH_TP_VM10() asm volatile("s_waitcnt vmcnt(10)" ::: "memory")
#define GH_TP_BAR() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define GH_TP_M(RH_, CH_) do { __builtin_amdgcn_s_setprio(1); GH_TP_QUAD(RH_, CH_) __builtin_amdgcn_s_setprio(0); GH_TP_BAR(); } while (0)
  // column sums (waves of the tiles that carry a bias gradient only): two more MFMAs in the first MFMA section that uses the row half.
  // (Round 6, measured and replaced: 32 unpack-adds per half-tile on the VALU in the load sections of the two light phases -- a dependent
  //  add chain as long as the other wave row's MFMA section; a single 2304 x 768 output with its bias gradient 198 -> 225 us.  Picking the
  //  tile by a run-time index into aF[] instead of this wave-uniform branch chain put the fragment array into scratch: 5 x slower.)
#define GH_TP_CS1(RH_, MI_) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                     \
    acs[RH_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, GH_TP_FRAG(aF[MI_][kk]), acs[RH_], 0, 0, 0);
#define GH_TP_CSUM(RH_) do { if ((RH_) == 0 ? do_cs0 : do_cs1) { if (wn == 0) { GH_TP_CS1(RH_, 0) } else if (wn == 1) { GH_TP_CS1(RH_, 1) }               \
                                          else if (wn == 2) { GH_TP_CS1(RH_, 2) } else { GH_TP_CS1(RH_, 3) } } } while (0)
#define GH_TP_MCS(RH_, CH_) do { __builtin_amdgcn_s_setprio(1); GH_TP_QUAD(RH_, CH_) GH_TP_CSUM(RH_); __builtin_amdgcn_s_setprio(0); GH_TP_BAR(); } while (0)

  // ---- prologue: tile 0 entirely, tile 1 except its second A half (phase 1 of tile 0 stages that one)
  stage(0, OPA, H0); stage(0, OPB, H0); stage(0, OPB, H1); stage(0, OPA, H1);
  stage(1, OPA, H0); stage(1, OPB, H0); stage(1, OPB, H1);
  GH_TP_VM10();
  asm volatile("s_barrier" ::: "memory");
  if (wm == 1) asm volatile("s_barrier" ::: "memory");      // the second wave row runs one barrier interval behind
  __builtin_amdgcn_sched_barrier(0);

#define GH_TP_TILE(B_)                                                                                                             \
  /* phase 1 */ GH_TP_RD_B(B_, 0) GH_TP_RD_A(B_, 0) stage(t + 1, OPA, H1); GH_TP_VM10(); GH_TP_WAIT_B(0); GH_TP_WAIT_A(); GH_TP_BAR(); GH_TP_MCS(0, 0); \
  /* phase 2 */ GH_TP_RD_B(B_, 1) stage(t + 2, OPA, H0); GH_TP_VM10(); GH_TP_WAIT_B(1); GH_TP_BAR(); GH_TP_M(0, 1);                   \
  /* phase 3 */ GH_TP_RD_A(B_, 1) stage(t + 2, OPB, H0); GH_TP_WAIT_A(); GH_TP_BAR(); GH_TP_MCS(1, 1);                             \
  /* phase 4 */ stage(t + 2, OPB, H1); GH_TP_VM10(); GH_TP_BAR(); GH_TP_M(1, 0);

  {
    int t = 0;
    for (; t + 1 < T; t += 2) {
      GH_TP_TILE(0)
      ++t;
      GH_TP_TILE(1)
      --t;
    }
    if (t < T) { GH_TP_TILE(0) }
  }
  if (wm == 0) asm volatile("s_barrier" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef GH_TP_TR
#undef GH_TP_RD_A
#undef GH_TP_RD_B
#undef GH_TP_FRAG
#undef GH_TP_QUAD
#undef GH_TP_WAIT_A
#undef GH_TP_WAIT_B
#undef GH_TP_VM10
#undef GH_TP_BAR
#undef GH_TP_M
#undef GH_TP_CS1
#undef GH_TP_CSUM
#undef GH_TP_MCS
#undef GH_TP_TILE

  if (GH_DBG_BITS(L) & 1) {      // (tool build: K loop only)
    float s = acs[0][0] + acs[1][0];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) s += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
    if (s == 12345.678f) P.C[0] = 0.f;
    return;
  }
  // every element of lane (l15, q)'s result is the whole sum of column l15 of tile mi = wn
#pragma unroll
  for (int rh = 0; rh < 2; ++rh) {
    const int c = i0 + rh * 128 + wm * 64 + wn * 16 + l15;
    if ((rh == 0 ? do_cs0 : do_cs1) && q == 0 && c < I) colsum[(size_t)ks * (size_t)P.colsum_stride + c] = acs[rh][0];
  }
  // partial tile: plain 16-byte stores (four consecutive columns per lane), summed by reduce_partials_kernel
  const int ldc = P.ldc;
  float* const C = P.C + (size_t)ks * (size_t)P.split_stride;
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int row = i0 + (mi >> 2) * 128 + wm * 64 + (mi & 3) * 16 + l15;
    if (row < I) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int col = j0 + (ni >> 1) * 128 + wn * 32 + (ni & 1) * 16 + 4 * q;
        if (col < J) __builtin_nontemporal_store(acc[mi][ni], reinterpret_cast<f32x4*>(C + (size_t)row * ldc + col));
      }
    }
  }
#endif
}

}  // namespace gh
