// Fast path of the grouped fp32 MFMA GEMM (same problem descriptors as gemm.hip.h).
//
// Preconditions (checked by the host, otherwise the generic kernel in gemm.hip.h runs):
//   every operand row is 16-byte aligned and its inner extent a multiple of 4 floats (float4
//   loads/stores everywhere), no row gather on the k-rows of B.
// What differs from the generic kernel:
//   * all addressing is hoisted out of the K loop: per-thread 32-bit element offsets, row validity
//     and (NT mode) the embedding-gather row ids are computed once; a K tile is NA+NB unconditional
//     global_load_dwordx4 from clamped addresses followed by selects -- no branches, no descriptor
//     re-loads, no dependent loads inside the loop, so the next tile's loads fly under the MFMAs;
//   * the MFMA operands are swapped (acc = mfma(b, a)), so a lane owns 4 CONSECUTIVE COLUMNS of one
//     output row: every epilogue access (bias, gate inputs, outputs) is one 16-byte vector op.
#pragma once
#include "gemm.hip.h"

namespace gh {

// Tile = (16*MI*WM) x (16*NI*WN): WM x WN waves, each owning MI x NI MFMA tiles of 16x16.
// BF: operands are rounded to bf16 when they are staged in LDS and multiplied with
// v_mfma_f32_16x16x16_bf16 (fp32 accumulate, everything outside the tile stays fp32) -- the opt-in path for
// BASELINE configs[4] ("h=768 bf16 ... MFMA projections"); the default and every parity claim are fp32.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));     // v_cvt_pk_bf16_f32 (RNE)
}

template <int WM, int WN, int NI, bool TN, int MI = 2, bool BF = false>
__global__ void __launch_bounds__(WM * WN * 64, 2)
gemm_fast_kernel(const Launch L_byval) {

  (void)L_byval;
  const GH_KARG Launch& L = *(const GH_KARG Launch*)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int NTHR = WM * WN * 64;
  constexpr int BM = 16 * MI * WM, BN = 16 * NI * WN, BK = 16;
  // LDS pitches == 4 (mod 32) dwords: with the k-rows of one MFMA step taken as {s, s+4, s+8, s+12}
  // the two 16-lane halves of a ds_read_b32 group sit 16 banks apart (conflict-free fragment reads),
  // and rows stay 16-byte aligned so the B tile is written with ds_write_b128.
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int A4 = BM * 4, B4 = BK * (BN / 4);
  constexpr int NA = (A4 + NTHR - 1) / NTHR, NB = (B4 + NTHR - 1) / NTHR;
  __shared__ float smem[2 * BK * LDA + 2 * BK * LDB];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA;
  // BF layout: [row][16 k as bf16] resp. [col][16 k as bf16] with a 40-byte pitch (10 dwords: the 8-byte fragment
  // reads of a half wave spread over all 32 banks exactly twice = the 2-cycle optimum); one MFMA covers the K tile
  constexpr int PB8 = 40;
  unsigned char* As8 = reinterpret_cast<unsigned char*>(smem);
  unsigned char* Bs8 = As8 + 2 * BM * PB8;
  static_assert(!BF || (2 * BM * PB8 + 2 * BN * PB8 <= (int)sizeof(float) * (2 * BK * LDA + 2 * BK * LDB)), "LDS");
  constexpr int BF4 = BN * 4;                                    // BF: B slots = (column, group of 4 k)
  constexpr int NBF = (BF4 + NTHR - 1) / NTHR;
  static_assert(!BF || NBF <= NB, "the bf16 B loader reuses the fp32 prefetch registers");

  // NT launches may also be split over K (few-row GEMMs): inner = (k-chunk, problem)
  const int n_inner = TN ? L.m_tiles * L.nprob : L.nprob * L.ksplit;
  const int n_outer = TN ? L.ksplit : L.m_tiles;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int outer = xcd + 8 * (slot / n_inner);
  const int inner = slot % n_inner;
  if (outer >= n_outer) return;
  const int prob = inner % L.nprob;
  const int m_tile = TN ? inner / L.nprob : outer;
  const int ks = TN ? outer : inner / L.nprob;
  const bool split = TN || L.ksplit > 1;
  const GH_KARG Problem& P = L.p[prob];
  const int M = P.M, N = P.N;
  const int m0 = m_tile * BM;
  if (m0 >= M) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int wrow = wm * 16 * MI, wcol = wn * 16 * NI;
  const int l15 = lane & 15, q = lane >> 4;

  // ---- loop-invariant operand description, copied to registers once
  const int nseg = TN ? 1 : P.nseg;
  const float* A0 = P.seg[0].A; const float* B0 = P.seg[0].B;
  const int lda0 = P.seg[0].lda, ldb0 = P.seg[0].ldb, K0 = P.seg[0].K;
  const float* A1 = nseg > 1 ? P.seg[1].A : A0; const float* B1 = nseg > 1 ? P.seg[1].B : B0;
  const int lda1 = nseg > 1 ? P.seg[1].lda : lda0, ldb1 = nseg > 1 ? P.seg[1].ldb : ldb0;
  const int K1 = nseg > 1 ? P.seg[1].K : 0;

  int kbeg = 0, kend = K0;
  if (split) {
    kbeg = ks * L.kchunk;
    kend = min(K0, kbeg + L.kchunk);
    if (kbeg >= kend) return;
  }
  const int nt0 = (kend - kbeg + BK - 1) / BK;
  // row tiles that lie entirely at or beyond seg0_rows have an all-zero segment-0 operand (the aggregation of
  // padding nodes in the node-compact layout): start at the first tile of segment 1
  const int toff = (!TN && nseg > 1 && P.seg0_rows > 0 && m0 >= P.seg0_rows) ? nt0 : 0;
  const int T = nt0 + (nseg > 1 ? (K1 + BK - 1) / BK : 0) - toff;

  // per-thread A slots
  unsigned a_off0[NA], a_off1[NA];
  bool a_ok[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int idx = tid + j * NTHR;
    if (!TN) {
      const int row = idx >> 2, gm = m0 + row;
      a_ok[j] = (idx < A4) && (gm < M);
      const int gmc = min(gm, M - 1);
      int s0 = gmc, s1 = gmc;
      if (P.seg[0].gatherA) s0 = P.seg[0].gatherA[gmc];          // embedding row id, read once per row
      if (nseg > 1 && P.seg[1].gatherA) s1 = P.seg[1].gatherA[gmc];
      a_off0[j] = a_ok[j] ? (unsigned)s0 * (unsigned)lda0 * 4u : 0x80000000u;      // BYTE offsets; invalid rows read zeros
      a_off1[j] = a_ok[j] ? (unsigned)s1 * (unsigned)lda1 * 4u : 0x80000000u;
    } else {
      const int c = m0 + 4 * (idx % (BM / 4));
      a_ok[j] = (idx < A4) && (c < M);
      a_off0[j] = (unsigned)min(c, M - 4);      // column offset; the k-row term is added per tile
      a_off1[j] = 0;
    }
  }
  // TN mode loads through buffer descriptors: `buffer_load_dwordx4 v, voffset, rsrc, soffset offen` with a per-thread
  // BYTE voffset fixed for the whole K loop (k-row within the tile, column) and a wave-uniform soffset that advances
  // with the K tile -- no address VALU per tile; rows beyond the K chunk and invalid columns (OOB marker) come back
  // as 0 from the hardware range check (record count = kend * ld * 4 bytes), so there are no selects either
  // (TN GEMM 86.8 -> 93.0 TF).  In NT mode the same scheme costs more than it saves: selecting the segment's
  // descriptor pushes the kernel over its SGPR budget (scratch traffic in the loop, 91 vs 98 TF), so NT keeps
  // clamped global loads.
  typedef __amdgpu_buffer_rsrc_t rsrc_t;
  constexpr unsigned OOB = 0x80000000u;
  unsigned ta_vo[NA], tb_vo[NB];
  if (!TN && !BF) {          // NT B operand through a per-tile descriptor (both K segments share ldb -- host-checked)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int idx = tid + j * NTHR;
      const int c = 4 * (idx % (BN / 4));
      tb_vo[j] = ((idx < B4) && (c < N)) ? ((unsigned)(idx / (BN / 4)) * (unsigned)ldb0 + (unsigned)c) * 4u : OOB;
    }
  }
  if (TN) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int idx = tid + j * NTHR;
      const int c = m0 + 4 * (idx % (BM / 4));
      ta_vo[j] = ((idx < A4) && (c < M)) ? ((unsigned)(idx / (BM / 4)) * (unsigned)lda0 + (unsigned)c) * 4u : OOB;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int idx = tid + j * NTHR;
      const int c = 4 * (idx % (BN / 4));
      tb_vo[j] = ((idx < B4) && (c < N)) ? ((unsigned)(idx / (BN / 4)) * (unsigned)ldb0 + (unsigned)c) * 4u : OOB;
    }
  }
  // per-thread B slots: column offset and validity (row term added per tile)
  unsigned b_col[NB];
  bool b_ok[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int idx = tid + j * NTHR;
    const int c = 4 * (idx % (BN / 4));
    b_ok[j] = (idx < B4) && (c < N);
    b_col[j] = (unsigned)min(c, N - 4);
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int mi_cnt = min(MI, (M - m0 - wrow + 15) / 16);
  const int ni_cnt = min(NI, (N - wcol + 15) / 16);

  float4 ra[NA], rb[NB];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 csum = zero4;                          // TN: running column sums of this thread's A slots (bias gradient)
  float* const colsum = TN ? P.colsum : nullptr;
  const int drop_mode = P.drop_mode;
  const unsigned drop_seed = P.drop_seed, drop_thresh = P.drop_thresh;
  const float drop_scale = P.drop_scale;
  const int drop_ld = P.drop_ld, drop_col0 = P.drop_col0;

  // Tile addressing shared by the load (issue) and store (mask + LDS write) halves.
  struct TileAddr { const float* Ab; const float* Bb; int ldb, k0, klim; bool s1; };
  auto tile_addr = [&](int t) __attribute__((always_inline)) {
    TileAddr a;
    t += toff;
    a.s1 = (!TN) && (t >= nt0);
    // blended with bit operations on purpose: with plain selects the compiler parks the two candidates in a scratch
    // array and indexes it inside the loop
    const unsigned long long m64 = 0ull - (unsigned long long)(a.s1 ? 1 : 0);
    const unsigned m32 = (unsigned)m64;
    a.Ab = (const float*)((unsigned long long)A0 ^ (((unsigned long long)A0 ^ (unsigned long long)A1) & m64));
    a.Bb = (const float*)((unsigned long long)B0 ^ (((unsigned long long)B0 ^ (unsigned long long)B1) & m64));
    a.ldb = (int)((unsigned)ldb0 ^ (((unsigned)ldb0 ^ (unsigned)ldb1) & m32));
    a.k0 = a.s1 ? (t - nt0) * BK : kbeg + t * BK;
    a.klim = (int)((unsigned)kend ^ (((unsigned)kend ^ (unsigned)K1) & m32));
    return a;
  };

  // issue only: raw 16-byte loads from clamped (always legal) addresses; nothing consumes them here,
  // so they stay in flight under the MFMAs of the current tile
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    const TileAddr a = tile_addr(t);
    if (TN && BF) {
      // both operands are k-major ([k][i], [k][j]); a lane needs 4 consecutive k of one column: four coalesced dword
      // loads per slot (64 lanes = 64 consecutive columns of one k row), clamped addresses, masked when stored
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * NTHR;
        const int kq = min(idx / BM, 3), col = min(m0 + idx % BM, M - 1);
        const float* ap = A0 + col;
        float4 v;
        v.x = ap[(unsigned)min(a.k0 + 4 * kq + 0, a.klim - 1) * (unsigned)lda0];
        v.y = ap[(unsigned)min(a.k0 + 4 * kq + 1, a.klim - 1) * (unsigned)lda0];
        v.z = ap[(unsigned)min(a.k0 + 4 * kq + 2, a.klim - 1) * (unsigned)lda0];
        v.w = ap[(unsigned)min(a.k0 + 4 * kq + 3, a.klim - 1) * (unsigned)lda0];
        ra[j] = v;
      }
    } else if (TN) {
      const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A0, 0, kend * lda0 * 4, 0x00020000);
      const rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B0, 0, kend * ldb0 * 4, 0x00020000);
#pragma unroll
      for (int j = 0; j < NA; ++j)
        ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rA, ta_vo[j], a.k0 * lda0 * 4, 0));
#pragma unroll
      for (int j = 0; j < NB; ++j)
        rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rB, tb_vo[j], a.k0 * ldb0 * 4, 0));
      return;
    }
    if (!TN) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int idx = tid + j * NTHR;
      // rows may be gathered (embedding table), so the record count is generous; invalid rows carry the OOB marker,
      // the k tail is clamped here and masked when stored
      const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.Ab, 0, (int)0x7fffffff, 0x00020000);
      const unsigned m32a = 0u - (unsigned)(a.s1 ? 1 : 0);
      const unsigned rowoff = a_off0[j] ^ ((a_off0[j] ^ a_off1[j]) & m32a);
      const unsigned vo = rowoff + 4u * (unsigned)min(a.k0 + 4 * (idx & 3), a.klim - 4);
      ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rA, vo, 0, 0));
    }
    }
    if (BF) {
      // B is k-major in memory but the bf16 MFMA wants 4 consecutive k per lane: four coalesced dword loads (64 lanes
      // = 64 consecutive columns of one k row) per slot instead of one 16-byte load along n
#pragma unroll
      for (int j = 0; j < NBF; ++j) {
        const int idx = tid + j * NTHR;
        const int kq = min(idx / BN, 3), n = min(idx % BN, N - 1);
        const float* bp = a.Bb + n;
        float4 v;
        v.x = bp[(unsigned)min(a.k0 + 4 * kq + 0, a.klim - 1) * (unsigned)a.ldb];
        v.y = bp[(unsigned)min(a.k0 + 4 * kq + 1, a.klim - 1) * (unsigned)a.ldb];
        v.z = bp[(unsigned)min(a.k0 + 4 * kq + 2, a.klim - 1) * (unsigned)a.ldb];
        v.w = bp[(unsigned)min(a.k0 + 4 * kq + 3, a.klim - 1) * (unsigned)a.ldb];
        rb[j] = v;
      }
      return;
    }
    {
      const rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.Bb, 0, a.klim * ldb0 * 4, 0x00020000);
      const int so = a.k0 * ldb0 * 4;
#pragma unroll
      for (int j = 0; j < NB; ++j)
        rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rB, tb_vo[j], so, 0));
    }
  };

  // mask out-of-range rows/columns/k (select, no branch) and write the k-major LDS image
  auto store_tile = [&](int buf, int t) __attribute__((always_inline)) {
    const TileAddr a = tile_addr(t);
    float* as = As + buf * BK * LDA;
    float* bs = Bs + buf * BK * LDB;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int idx = tid + j * NTHR;
      if (idx < A4) {
        if (!TN) {
          const int row = idx >> 2, kq = idx & 3;
          const bool ok = a.k0 + 4 * kq < a.klim;          // invalid rows already came back as zeros
          float4 v = ok ? ra[j] : zero4;
          if (drop_mode == 1 && !a.s1)
            v = drop4(v, drop_seed, (unsigned)(m0 + row) * (unsigned)drop_ld + (unsigned)(a.k0 + 4 * kq), drop_thresh, drop_scale);
          if (BF) {
            *reinterpret_cast<uint2*>(As8 + (buf * BM + row) * PB8 + kq * 8) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
          } else {
            as[(4 * kq + 0) * LDA + row] = v.x;
            as[(4 * kq + 1) * LDA + row] = v.y;
            as[(4 * kq + 2) * LDA + row] = v.z;
            as[(4 * kq + 3) * LDA + row] = v.w;
          }
        } else if (BF) {
          const int kq = idx / BM, i = idx % BM;
          const bool iok = m0 + i < M;
          float4 v = ra[j];
          v.x = (iok && a.k0 + 4 * kq + 0 < a.klim) ? v.x : 0.f;
          v.y = (iok && a.k0 + 4 * kq + 1 < a.klim) ? v.y : 0.f;
          v.z = (iok && a.k0 + 4 * kq + 2 < a.klim) ? v.z : 0.f;
          v.w = (iok && a.k0 + 4 * kq + 3 < a.klim) ? v.w : 0.f;
          *reinterpret_cast<uint2*>(As8 + (buf * BM + i) * PB8 + kq * 8) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
        } else {
          const int krow = idx / (BM / 4), c = 4 * (idx % (BM / 4));
          *reinterpret_cast<float4*>(as + krow * LDA + c) = ra[j];       // already zero where out of range
          if (colsum) { csum.x += ra[j].x; csum.y += ra[j].y; csum.z += ra[j].z; csum.w += ra[j].w; }
        }
      }
    }
    if (BF) {
#pragma unroll
      for (int j = 0; j < NBF; ++j) {
        const int idx = tid + j * NTHR;
        if (idx < BF4) {
          const int kq = idx / BN, n = idx % BN;
          const bool nok = n < N;
          float4 v = rb[j];
          v.x = (nok && a.k0 + 4 * kq + 0 < a.klim) ? v.x : 0.f;
          v.y = (nok && a.k0 + 4 * kq + 1 < a.klim) ? v.y : 0.f;
          v.z = (nok && a.k0 + 4 * kq + 2 < a.klim) ? v.z : 0.f;
          v.w = (nok && a.k0 + 4 * kq + 3 < a.klim) ? v.w : 0.f;
          *reinterpret_cast<uint2*>(Bs8 + (buf * BN + n) * PB8 + kq * 8) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int idx = tid + j * NTHR;
      if (idx < B4) {
        const int krow = idx / (BN / 4), c = 4 * (idx % (BN / 4));
        const bool ok = true;            // out-of-range rows / columns were zero-filled by the buffer range check
        const float4 v = ok ? rb[j] : zero4;
        *reinterpret_cast<float4*>(bs + krow * LDB + c) = v;
      }
    }
  };

  // swapped operands: acc[mi][ni][r] = C[row = wrow + mi*16 + l15][col = wcol + ni*16 + 4*q + r]
  // NV = number of valid 16-wide column tiles of this wave when known at compile time (NI or NI-1:
  // N = 300 leaves the last wave one tile short), -1 = guarded per tile (ragged M / N tails).
  auto compute = [&](int buf, auto NVT, auto HALFT) __attribute__((always_inline)) {
    constexpr int NV = decltype(NVT)::value;
    constexpr int S0 = decltype(HALFT)::value * 2;      // k-steps {0,1} or {2,3} of the tile
    if constexpr (BF) {
      // one MFMA spans the whole 16-deep K tile: the first half of the tile does the lower MI/2 row tiles, the
      // second half the upper ones (the LDS refill still sits between them)
      constexpr int H = decltype(HALFT)::value;
      const unsigned char* as8 = As8 + (buf * BM + wrow + l15) * PB8 + q * 8;
      const unsigned char* bs8 = Bs8 + (buf * BN + wcol + l15) * PB8 + q * 8;
      s16x4_t bfr[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        if (NV < 0 || ni < NV) bfr[ni] = __builtin_bit_cast(s16x4_t, *reinterpret_cast<const uint2*>(bs8 + ni * 16 * PB8));
#pragma unroll
      for (int mi = H * (MI / 2); mi < (H + 1) * (MI / 2); ++mi) {
        const s16x4_t afr = __builtin_bit_cast(s16x4_t, *reinterpret_cast<const uint2*>(as8 + mi * 16 * PB8));
        if (NV >= 0 || mi < mi_cnt) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            if ((NV >= 0 && ni < NV) || (NV < 0 && ni < ni_cnt))
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bfr[ni], afr, acc[mi][ni], 0, 0, 0);
        }
      }
      return;
    }
    const float* as = As + buf * BK * LDA + wrow + l15;
    const float* bs = Bs + buf * BK * LDB + wcol + l15;
    const int kq = 4 * q;
#pragma unroll
    for (int s = S0; s < S0 + 2; ++s) {
      const int kr = s + kq;
      float a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = as[kr * LDA + mi * 16];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        if (NV < 0 || ni < NV) b[ni] = bs[kr * LDB + ni * 16];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        if (NV >= 0 || mi < mi_cnt) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            if ((NV >= 0 && ni < NV) || (NV < 0 && ni < ni_cnt))
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
        }
      }
    }
  };
  const int path = (mi_cnt == MI && ni_cnt == NI) ? 0 : ((mi_cnt == MI && ni_cnt == NI - 1) ? 1 : 2);

  // Software pipeline, one barrier per K tile:
  //   registers hold tile t+1 (loaded one iteration ago, so HBM/L2 latency has a whole tile to hide);
  //   halfway through the MFMAs of tile t they are written to the other LDS buffer and the loads of
  //   tile t+2 are issued, so the only thing left at the end of the tile is the barrier.
  load_tile(0);
  store_tile(0, 0);
  if (T > 1) load_tile(1);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    if (path == 0) compute(t & 1, std::integral_constant<int, NI>{}, std::integral_constant<int, 0>{});
    else if (path == 1) compute(t & 1, std::integral_constant<int, NI - 1>{}, std::integral_constant<int, 0>{});
    else compute(t & 1, std::integral_constant<int, -1>{}, std::integral_constant<int, 0>{});
    if (t + 1 < T) store_tile((t + 1) & 1, t + 1);
    if (t + 2 < T) load_tile(t + 2);
    if (path == 0) compute(t & 1, std::integral_constant<int, NI>{}, std::integral_constant<int, 1>{});
    else if (path == 1) compute(t & 1, std::integral_constant<int, NI - 1>{}, std::integral_constant<int, 1>{});
    else compute(t & 1, std::integral_constant<int, -1>{}, std::integral_constant<int, 1>{});
    __syncthreads();
  }

  if (TN && colsum) {
    // every A tile passed through store_tile exactly once: fold the 16 k-rows of each 4-column group (LDS is free
    // after the loop's last barrier) and write this K chunk's partial bias gradient
    constexpr int CG = BM / 4;
    float4* red = reinterpret_cast<float4*>(smem);
    if (tid < A4) red[tid] = csum;
    __syncthreads();
    if (tid < CG) {
      float4 v = zero4;
#pragma unroll
      for (int r = 0; r < A4 / CG; ++r) { const float4 x = red[r * CG + tid]; v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w; }
      const int c = m0 + 4 * tid;
      if (c < M) *reinterpret_cast<float4*>(colsum + (size_t)ks * (size_t)P.colsum_stride + c) = v;
    }
  }

  // -------------------------------------------------------------------- epilogue (16-byte vector accesses)
  const int epi = P.epi;
  const int ldc = P.ldc;
  float* const C = P.C + (split ? (size_t)ks * (size_t)P.split_stride : (size_t)0);
  if (epi == EPI_ATT) {
    float* red = smem;   // [WN][BM][8]
    const int heads = P.heads;
    const float* U = P.u; const float* W2 = P.w2; float* E = P.e;
    const int ldu = P.ldu, R = P.R;
    const int32_t* rowg = P.rowg;
    auto att_rows = [&](auto MIT) __attribute__((always_inline)) {
      constexpr int mi = decltype(MIT)::value;
      const int lrow = wrow + mi * 16 + l15;
      const int row = m0 + lrow;
      float pe[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) pe[c] = 0.f;
      if (mi < mi_cnt && row < M) {
        const float* urow = U + (size_t)(rowg ? rowg[row] : row / R) * ldu;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int col = wcol + ni * 16 + 4 * q;
          if (ni < ni_cnt && col < N) {
            const float4 u4 = *reinterpret_cast<const float4*>(urow + col);
            float4 t;
            t.x = tanhf_(acc[mi][ni][0] + u4.x); t.y = tanhf_(acc[mi][ni][1] + u4.y);
            t.z = tanhf_(acc[mi][ni][2] + u4.z); t.w = tanhf_(acc[mi][ni][3] + u4.w);
            *reinterpret_cast<float4*>(C + (size_t)row * ldc + col) = t;
#pragma unroll
            for (int c = 0; c < 8; ++c)
              if (c < heads) {
                const float4 w = *reinterpret_cast<const float4*>(W2 + (size_t)c * N + col);
                pe[c] += t.x * w.x + t.y * w.y + t.z * w.z + t.w * w.w;
              }
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float v = pe[c];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        pe[c] = v;
      }
      if (WN == 1) {
        if (q == 0 && row < M && mi < mi_cnt)
          for (int c = 0; c < heads; ++c) E[(size_t)row * heads + c] = pe[c];
      } else {
        if (q == 0)
          for (int c = 0; c < 8; ++c) red[(wn * BM + lrow) * 8 + c] = pe[c];
      }
    };
    att_rows(std::integral_constant<int, 0>{});
    att_rows(std::integral_constant<int, 1>{});
    if constexpr (MI > 2) { att_rows(std::integral_constant<int, 2>{}); att_rows(std::integral_constant<int, 3>{}); }
    static_assert(MI == 2 || MI == 4, "MI is 2 or 4");
    if (WN > 1) {
      __syncthreads();
      for (int i = tid; i < BM * 8; i += NTHR) {
        const int lrow = i >> 3, c = i & 7, row = m0 + lrow;
        if (row < M && c < heads) {
          float v = 0.f;
          for (int w = 0; w < WN; ++w) v += red[(w * BM + lrow) * 8 + c];
          E[(size_t)row * heads + c] = v;
        }
      }
    }
    return;
  }

  const float* bias = P.bias;
  float* out1 = P.out1;
  const float* in0 = P.in0;
  const float* in1 = P.in1;
  const int accumulate = P.accumulate;
  auto epi_rows = [&](auto MIT) __attribute__((always_inline)) {
    constexpr int mi = decltype(MIT)::value;
    const int row = m0 + wrow + mi * 16 + l15;
    const bool row_ok = (mi < mi_cnt) && (row < M);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = wcol + ni * 16 + 4 * q;
      if (row_ok && ni < ni_cnt && col < N) {
        const size_t o = (size_t)row * ldc + col;
        float4 v = make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
        if (bias) {
          const float4 b4 = *reinterpret_cast<const float4*>(bias + col);
          v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
        }
        if (epi == EPI_STORE) {
          if (drop_mode == 3)
            v = drop4(v, drop_seed, (unsigned)row * (unsigned)drop_ld + (unsigned)(drop_col0 + col), drop_thresh, drop_scale);
          if (accumulate) {
            const float4 p = *reinterpret_cast<const float4*>(C + o);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
          }
          *reinterpret_cast<float4*>(C + o) = v;
        } else if (epi == EPI_SIGMOID_Z) {
          *reinterpret_cast<float4*>(C + o) = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
        } else if (epi == EPI_SIGMOID_R) {
          const float4 r = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
          const float4 x = *reinterpret_cast<const float4*>(in0 + o);
          *reinterpret_cast<float4*>(C + o) = r;
          *reinterpret_cast<float4*>(out1 + o) = make_float4(r.x * x.x, r.y * x.y, r.z * x.z, r.w * x.w);
        } else if (epi == EPI_TANH_H) {
          const float4 h = make_float4(tanhf_(v.x), tanhf_(v.y), tanhf_(v.z), tanhf_(v.w));
          const float4 z = *reinterpret_cast<const float4*>(in0 + o);
          const float4 x = *reinterpret_cast<const float4*>(in1 + o);
          *reinterpret_cast<float4*>(C + o) = h;
          *reinterpret_cast<float4*>(out1 + o) =
              make_float4(h.x * z.x + x.x * (1.f - z.x), h.y * z.y + x.y * (1.f - z.y),
                          h.z * z.z + x.z * (1.f - z.z), h.w * z.w + x.w * (1.f - z.w));
        } else if (epi == EPI_BWD_DRX) {
          const float4 x = *reinterpret_cast<const float4*>(in0 + o);
          const float4 r = *reinterpret_cast<const float4*>(in1 + o);
          float4 d = *reinterpret_cast<const float4*>(out1 + o);
          *reinterpret_cast<float4*>(C + o) =
              make_float4(v.x * x.x * r.x * (1.f - r.x), v.y * x.y * r.y * (1.f - r.y),
                          v.z * x.z * r.z * (1.f - r.z), v.w * x.w * r.w * (1.f - r.w));
          d.x += v.x * r.x; d.y += v.y * r.y; d.z += v.z * r.z; d.w += v.w * r.w;
          *reinterpret_cast<float4*>(out1 + o) = d;
        } else if (epi == EPI_ATOMIC) {
          atomicAdd(C + o + 0, v.x); atomicAdd(C + o + 1, v.y);
          atomicAdd(C + o + 2, v.z); atomicAdd(C + o + 3, v.w);
        }
      }
    }
  };
  epi_rows(std::integral_constant<int, 0>{});
  epi_rows(std::integral_constant<int, 1>{});
  if constexpr (MI > 2) { epi_rows(std::integral_constant<int, 2>{}); epi_rows(std::integral_constant<int, 3>{}); }
}

}  // namespace gh
