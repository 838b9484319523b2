// Grouped fp32 MFMA GEMM for gfx950 (v_mfma_f32_16x16x4_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// One launch runs up to GH_MAX_PROBLEMS problems that share their row space, each
//     C_p[M][N_p] = epilogue_p( sum_seg  A_seg[M][K_seg] . B_seg[N_p][K_seg]^T )
// with up to two K-segments (so [a | x] . [W0 | W1]^T needs no concatenated copy); NT mode takes B as
// [N][ldb] (the weight as PyTorch stores it; dX products take the cached transpose).  This file is the generic,
// scalar-guarded kernel (odd shapes, unaligned operands); the fast paths are gemm_nt.hip.h (NT) and
// gemm_tn.hip.h (TN).  TN mode (weight gradients) computes
//     C_p[I][J] += sum_m A[m][I]^T B[m][J]          (split over m, fp32 atomics).
//
// Tiling: workgroup = WM x WN waves, wave tile = 32 x (16*NI), K tile = 16.
//   (4,1,19): 128 x 304 -- the 96000-row activation GEMMs with N = 300 in ONE column block, so
//             every activation row is read from HBM once per GEMM.
//   (1,4,5) :  32 x 320 -- the few-hundred-row GEMMs of the evidence level / left branch.
// LDS image is k-major with row pitch == 2 (mod 32) dwords: with the k-rows of one MFMA step chosen
// as {s, s+8, s+4, s+12} both the transposing stores of A and the fragment reads of A and B are
// bank-conflict free (ds_read_b32 / ds_write_b32 group = 32 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace gh {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum EpiKind : int {
  EPI_STORE = 0,   // C = v (+bias[col]) (+C when accumulate)
  EPI_SIGMOID_Z,   // z = sigmoid(v+bias) -> C
  EPI_SIGMOID_R,   // r = sigmoid(v+bias) -> C ; out1 = r * in0            (in0 = xp)
  EPI_TANH_H,      // h = tanh(v+bias) -> C ; out1 = h*z + xp*(1-z)        (in0 = z, in1 = xp)
  EPI_ATT,         // t = tanh(v + u[rowg ? rowg[row] : row/R][col]) -> C ; e[row][c] = sum_col t*w2[c][col]
  EPI_BWD_DRX,     // v = d(r*xp): C(drp) = v*xp*r*(1-r) ; out1(dxp) += v*r  (in0 = xp, in1 = r)
  EPI_ATOMIC,      // atomicAdd(C, v)   (TN split-K)
  EPI_GATE_PRE,    // g = v (+ gin): the GGNN cell backward's elementwise head fused into the GEMM that PRODUCES g (fast NT
                   // kernel only): C(dhp) = g z (1-h^2) ; out1(dzp) = g (h-xp) z (1-z) ; out2(dxp) = g (1-z)
                   // (in0 = z, in1 = hh, in2 = xp of the cell that consumes g; g itself is never stored)
};

struct Seg {
  const float* A; const int32_t* gatherA; int lda; int vecA;
  const float* B; const int32_t* gatherB; int ldb; int vecB;
  int K;
};

struct Problem {
  Seg seg[2];
  int nseg;
  int M, N;
  int epi;
  int accumulate;
  float* C; int ldc;
  const float* bias;
  const float* bias2;       // optional second bias added to `bias` (b?0 + b?1 of a gate; NT fast kernel and the generic kernel)
  float* out1;
  const float* in0; const float* in1;   // same leading dimension as C
  const float* u; const float* w2; float* e; int ldu; int R; int heads;
  int seg0_rows;            // > 0: segment 0's A operand is all zero for rows >= seg0_rows (fast kernel skips it per tile)
  const int32_t* rowg;      // EPI_ATT: u row of output row m is rowg[m] when non-null (node-compact layout), else m / R
  long long split_stride;   // TN: element offset of K-chunk `ks`'s partial tile (0 = all chunks hit C, atomically)
  // TN: when non-null, the column sums of the A operand over this workgroup's K chunk (= the bias gradient that
  // belongs to this weight gradient) are written to colsum[ks * colsum_stride + column]; fast kernel only
  float* colsum; long long colsum_stride;
  // stateless input dropout (wrapper.py:189-190): element (row, col) of the dropped matrix [rows][drop_ld] is kept
  // iff hash(seed, row*drop_ld + col) >= drop_thresh and then scaled by drop_scale = 1/(1-p).
  // drop_mode: 0 off, 1 = A of segment 0 (NT), 3 = C in the EPI_STORE epilogue (gradient w.r.t. the dropped
  //            input).  Fast kernel only.  (The weight-gradient GEMM gets its masked operand from gather_rows:
  //            hashing in the TN loader pushed that kernel over 168 VGPRs, i.e. from 3 to 2 waves per SIMD.)
  int drop_mode; int drop_ld; int drop_col0; unsigned drop_seed; unsigned drop_thresh; float drop_scale;
  // bf16 storage pipeline (gemm_nt.hip.h MODE 2 / gemm_tn.hip.h bf16): elt = 1 -> A and B hold bf16 (lda/ldb/K in
  // elements); io bits say which epilogue streams are bf16: 1 = C, 2 = out1, 4 = in0, 8 = in1, 16 = in2, 32 = out2 (the last
  // two: EPI_GATE_PRE); c32 (EPI_TANH_H): also
  // write the fp32 value of out1 there (the cell output the fp32 consumers read)
  int elt; int io; float* c32;
  // EPI_TANH_H with the fused scorer projection on a COLUMN BLOCK of the row (narrow tile): this block's partial dot product is
  // ADDED to e[row] (zeroed by the caller; two blocks -> two addends -> the sum does not depend on their order)
  int e_atomic;
  // EPI_GATE_PRE: optional addend of g (same shape as C), third input stream, third output stream
  const float* gin; const float* in2; float* out2;
};

#define GH_MAX_PROBLEMS 12     // (sizeof(Launch) = 12 x 336 + 32 = 4064 of the 4096 kernarg bytes: the head's 3556-wide products are 12 column blocks)
struct Launch {
  Problem p[GH_MAX_PROBLEMS];
  int nprob;
  int m_tiles;   // max over problems of ceil(M / BM)
  int ksplit;    // TN: number of K chunks (1 otherwise)
  int kchunk;    // TN: rows per chunk, multiple of 16
  int dbg;       // measurement only (GH_DBG): bit 0 = skip the epilogue, bit 1 = skip the K loop
  int n_tiles;   // gemm_tn_pp_kernel: max over problems of ceil(N / 256) (its problems are whole outputs, not column blocks)
  int per;       // gemm_tn_pp_kernel: work items per XCD (XCD x runs items [x per, (x + 1) per) of the chunk-major list)
};
static_assert(sizeof(Launch) <= 4096, "Launch travels by value in the kernarg segment (4 KB)");

__host__ __device__ __forceinline__ unsigned drop_hash(unsigned seed, unsigned idx) {
  unsigned x = idx * 0x9E3779B1u + seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float4 drop4(float4 v, unsigned seed, unsigned idx, unsigned thresh, float scale) {
  v.x = drop_hash(seed, idx + 0) >= thresh ? v.x * scale : 0.f;
  v.y = drop_hash(seed, idx + 1) >= thresh ? v.y * scale : 0.f;
  v.z = drop_hash(seed, idx + 2) >= thresh ? v.z * scale : 0.f;
  v.w = drop_hash(seed, idx + 3) >= thresh ? v.w * scale : 0.f;
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); exact to ~1e-7 abs, saturates cleanly for |x| large
  float e = __expf(2.0f * x);
  return 1.0f - 2.0f / (e + 1.0f);
}

#define GH_KARG __attribute__((address_space(4)))

template <int WM, int WN, int NI, bool TN>
__global__ void __launch_bounds__(WM * WN * 64, 2)
gemm_kernel(const Launch L_byval) {
  // The descriptor table is indexed with a run-time problem id: read it straight from the kernarg
  // segment (constant address space, scalar loads) instead of letting the by-value copy be
  // spilled to scratch for dynamic indexing.
  (void)L_byval;
  const GH_KARG Launch& L = *(const GH_KARG Launch*)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int NTHR = WM * WN * 64;
  constexpr int BM = 32 * WM, BN = 16 * NI * WN, BK = 16;
  constexpr int LDA = BM + 2, LDB = BN + 2;
  constexpr int A4 = BM * 4;              // float4 per A tile
  constexpr int B4 = BK * (BN / 4);       // float4 per B tile
  constexpr int NA = (A4 + NTHR - 1) / NTHR;
  constexpr int NB = (B4 + NTHR - 1) / NTHR;
  __shared__ float smem[2 * BK * LDA + 2 * BK * LDB];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA;

  // ---- XCD-aware work decode: blocks b, b+8, b+16.. share an XCD (and its L2); give them the
  //      problems that re-read the same activation panel (NT/NN) or the same row chunk (TN).
  const int n_inner = TN ? L.m_tiles * L.nprob : L.nprob;
  const int n_outer = TN ? L.ksplit : L.m_tiles;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int outer = xcd + 8 * (slot / n_inner);
  const int inner = slot % n_inner;
  if (outer >= n_outer) return;
  const int prob = TN ? inner % L.nprob : inner;
  const int m_tile = TN ? inner / L.nprob : outer;
  const int ks = TN ? outer : 0;
  const GH_KARG Problem& P = L.p[prob];
  const int M = P.M, N = P.N;
  const int m0 = m_tile * BM;
  if (m0 >= M) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int wrow = wm * 32, wcol = wn * 16 * NI;
  const int l15 = lane & 15, q = lane >> 4;

  int nt0, T, kbeg = 0, kend = 0;
  if (TN) {
    kbeg = ks * L.kchunk;
    kend = min(P.seg[0].K, kbeg + L.kchunk);
    if (kbeg >= kend) return;
    nt0 = (kend - kbeg + BK - 1) / BK;
    T = nt0;
  } else {
    nt0 = (P.seg[0].K + BK - 1) / BK;
    T = nt0 + (P.nseg > 1 ? (P.seg[1].K + BK - 1) / BK : 0);
  }

  f32x4 acc[2][NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // wave-uniform validity of the 16-wide row/column tiles this wave owns (ragged M / N tails)
  const int mi_cnt = min(2, (M - m0 - wrow + 15) / 16);
  const int ni_cnt = min(NI, (N - wcol + 15) / 16);

  float4 ra[NA], rb[NB];

  auto load_tile = [&](int t) {
    const int si = (!TN && t >= nt0) ? 1 : 0;
    const GH_KARG Seg& S = P.seg[si];
    const int k0 = TN ? (kbeg + t * BK) : ((si ? t - nt0 : t) * BK);
    const int klim = TN ? kend : S.K;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int idx = tid + j * NTHR;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < A4) {
        if (!TN) {
          const int row = idx >> 2, k = k0 + 4 * (idx & 3), gm = m0 + row;
          if (gm < M && k < klim) {
            const int srow = S.gatherA ? S.gatherA[gm] : gm;
            const float* p = S.A + (size_t)srow * S.lda + k;
            if (S.vecA) v = *reinterpret_cast<const float4*>(p);
            else { v.x = p[0]; if (k + 1 < klim) v.y = p[1]; if (k + 2 < klim) v.z = p[2]; if (k + 3 < klim) v.w = p[3]; }
          }
        } else {
          const int krow = idx / (BM / 4), i = m0 + 4 * (idx % (BM / 4)), gk = k0 + krow;
          if (gk < klim && i < M) {
            const int srow = S.gatherA ? S.gatherA[gk] : gk;
            const float* p = S.A + (size_t)srow * S.lda + i;
            if (S.vecA) v = *reinterpret_cast<const float4*>(p);
            else { v.x = p[0]; if (i + 1 < M) v.y = p[1]; if (i + 2 < M) v.z = p[2]; if (i + 3 < M) v.w = p[3]; }
          }
        }
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int idx = tid + j * NTHR;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < B4) {
        const int krow = idx / (BN / 4), n = 4 * (idx % (BN / 4)), gk = k0 + krow;
        if (gk < klim && n < N) {
          if (!TN) {        // NT: B is [N][ldb], contraction-contiguous (the weight as stored); element (k, n) = B[n][k]
            const float* p = S.B + (size_t)n * S.ldb + gk;
            v.x = p[0];
            if (n + 1 < N) v.y = p[(size_t)S.ldb];
            if (n + 2 < N) v.z = p[2 * (size_t)S.ldb];
            if (n + 3 < N) v.w = p[3 * (size_t)S.ldb];
          } else {
            const int srow = S.gatherB ? S.gatherB[gk] : gk;
            const float* p = S.B + (size_t)srow * S.ldb + n;
            if (S.vecB) v = *reinterpret_cast<const float4*>(p);
            else { v.x = p[0]; if (n + 1 < N) v.y = p[1]; if (n + 2 < N) v.z = p[2]; if (n + 3 < N) v.w = p[3]; }
          }
        }
      }
      rb[j] = v;
    }
  };

  auto store_tile = [&](int buf) {
    float* as = As + buf * BK * LDA;
    float* bs = Bs + buf * BK * LDB;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int idx = tid + j * NTHR;
      if (idx < A4) {
        if (!TN) {
          const int row = idx >> 2, kq = idx & 3;
          as[(4 * kq + 0) * LDA + row] = ra[j].x;
          as[(4 * kq + 1) * LDA + row] = ra[j].y;
          as[(4 * kq + 2) * LDA + row] = ra[j].z;
          as[(4 * kq + 3) * LDA + row] = ra[j].w;
        } else {
          const int krow = idx / (BM / 4), c = 4 * (idx % (BM / 4));
          float2* d = reinterpret_cast<float2*>(as + krow * LDA + c);
          d[0] = make_float2(ra[j].x, ra[j].y);
          d[1] = make_float2(ra[j].z, ra[j].w);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int idx = tid + j * NTHR;
      if (idx < B4) {
        const int krow = idx / (BN / 4), c = 4 * (idx % (BN / 4));
        float2* d = reinterpret_cast<float2*>(bs + krow * LDB + c);
        d[0] = make_float2(rb[j].x, rb[j].y);
        d[1] = make_float2(rb[j].z, rb[j].w);
      }
    }
  };

  // FULL: every 16-wide tile of this wave is inside the matrix (the common case) -> branch-free MFMA stream
  auto compute = [&](int buf, auto FULLT) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(FULLT)::value;
    const float* as = As + buf * BK * LDA + wrow + l15;
    const float* bs = Bs + buf * BK * LDB + wcol + l15;
    const int kq = 8 * (q & 1) + 4 * (q >> 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int kr = s + kq;
      float a[2], b[NI];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = as[kr * LDA + mi * 16];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = bs[kr * LDB + ni * 16];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        if (FULL || mi < mi_cnt) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            if (FULL || ni < ni_cnt)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
      }
    }
  };
  const bool full = (mi_cnt == 2) && (ni_cnt == NI);

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T) load_tile(t + 1);
    if (full) compute(t & 1, std::true_type{});
    else compute(t & 1, std::false_type{});
    if (t + 1 < T) store_tile((t + 1) & 1);
    __syncthreads();
  }

  // -------------------------------------------------------------------- epilogue
  const int epi = P.epi;
  const int ldc = P.ldc;
  float* const Cout = P.C + (TN ? (size_t)ks * (size_t)P.split_stride : (size_t)0);
  if (epi == EPI_ATT) {
    // row-wise head scores need every column of the row: reduce over this lane's tiles, the 16
    // lanes of the row, and (WN > 1) the waves along N through LDS.
    float* red = smem;   // [WN][BM][8]
    const int heads = P.heads;
    auto att_rows = [&](auto MI) __attribute__((always_inline)) {
      constexpr int mi = decltype(MI)::value;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int lrow = wrow + mi * 16 + q * 4 + reg;
        const int row = m0 + lrow;
        float pe[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) pe[c] = 0.f;
        if (mi < mi_cnt && row < M) {
          const float* urow = P.u + (size_t)(P.rowg ? P.rowg[row] : row / P.R) * P.ldu;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const int col = wcol + ni * 16 + l15;
            if (ni < ni_cnt && col < N) {
              const float t = tanhf_(acc[mi][ni][reg] + urow[col]);
              P.C[(size_t)row * ldc + col] = t;
#pragma unroll
              for (int c = 0; c < 8; ++c)
                if (c < heads) pe[c] += t * P.w2[c * N + col];
            }
          }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float v = pe[c];
          v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
          pe[c] = v;
        }
        if (WN == 1) {
          if (l15 == 0 && row < M && mi < mi_cnt)
            for (int c = 0; c < heads; ++c) P.e[(size_t)row * heads + c] = pe[c];
        } else {
          if (l15 == 0)
            for (int c = 0; c < 8; ++c) red[(wn * BM + lrow) * 8 + c] = pe[c];
        }
      }
    };
    att_rows(std::integral_constant<int, 0>{});
    att_rows(std::integral_constant<int, 1>{});
    if (WN > 1) {
      __syncthreads();
      for (int i = tid; i < BM * 8; i += NTHR) {
        const int lrow = i >> 3, c = i & 7, row = m0 + lrow;
        if (row < M && c < heads) {
          float v = 0.f;
          for (int w = 0; w < WN; ++w) v += red[(w * BM + lrow) * 8 + c];
          P.e[(size_t)row * heads + c] = v;
        }
      }
    }
    return;
  }

  auto epi_rows = [&](auto MI) __attribute__((always_inline)) {
    constexpr int mi = decltype(MI)::value;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = wcol + ni * 16 + l15;
      const bool tile_ok = (mi < mi_cnt) && (ni < ni_cnt) && (col < N);
      const float bias = ((tile_ok && P.bias) ? P.bias[col] : 0.f) + ((tile_ok && P.bias2) ? P.bias2[col] : 0.f);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = m0 + wrow + mi * 16 + q * 4 + reg;
        if (tile_ok && row < M) {
          const size_t o = (size_t)row * ldc + col;
          const float v = acc[mi][ni][reg] + bias;
          if (epi == EPI_STORE) {
            Cout[o] = P.accumulate ? Cout[o] + v : v;
          } else if (epi == EPI_SIGMOID_Z) {
            Cout[o] = sigmoidf_(v);
          } else if (epi == EPI_SIGMOID_R) {
            const float r = sigmoidf_(v);
            Cout[o] = r;
            P.out1[o] = r * P.in0[o];
          } else if (epi == EPI_TANH_H) {
            const float h = tanhf_(v), z = P.in0[o], xp = P.in1[o];
            Cout[o] = h;
            P.out1[o] = h * z + xp * (1.f - z);
          } else if (epi == EPI_BWD_DRX) {
            const float xp = P.in0[o], r = P.in1[o];
            Cout[o] = v * xp * r * (1.f - r);
            P.out1[o] += v * r;
          } else if (epi == EPI_ATOMIC) {
            atomicAdd(Cout + o, v);
          }
        }
      }
    }
  };
  epi_rows(std::integral_constant<int, 0>{});
  epi_rows(std::integral_constant<int, 1>{});
}

// ---------------------------------------------------------------------------------- host side
inline int vec_ok(const void* p, int ld, int inner) {
  return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 4 == 0) && (inner % 4 == 0);
}

inline Seg make_seg(const float* A, int lda, const float* B, int ldb, int K, int n_inner,
                    const int32_t* gatherA = nullptr, const int32_t* gatherB = nullptr) {
  Seg s;
  s.A = A; s.lda = lda; s.gatherA = gatherA; s.vecA = vec_ok(A, lda, K);
  s.B = B; s.ldb = ldb; s.gatherB = gatherB; s.vecB = vec_ok(B, ldb, n_inner);
  s.K = K;
  return s;
}
// NT: A [M][lda] (rows optionally gathered), B [N][ldb]; both rows hold the K contraction values contiguously
inline Seg make_seg_nt(const float* A, int lda, const float* B, int ldb, int K, const int32_t* gatherA = nullptr, int elt = 0) {
  Seg s;
  const int q = elt ? 8 : 4;          // elements per 16 bytes
  s.A = A; s.lda = lda; s.gatherA = gatherA; s.vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && lda % q == 0 && K % q == 0;
  s.B = B; s.ldb = ldb; s.gatherB = nullptr; s.vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && ldb % q == 0 && K % q == 0;
  s.K = K;
  return s;
}
inline Seg make_seg_tn(const float* A, int lda, int I, const float* B, int ldb, int J, int K,
                       const int32_t* gatherA = nullptr, const int32_t* gatherB = nullptr) {
  Seg s;
  s.A = A; s.lda = lda; s.gatherA = gatherA; s.vecA = vec_ok(A, lda, I);
  s.B = B; s.ldb = ldb; s.gatherB = gatherB; s.vecB = vec_ok(B, ldb, J);
  s.K = K;
  return s;
}

inline Problem make_problem(int M, int N, int epi, float* C, int ldc) {
  Problem p;
  memset(&p, 0, sizeof(p));
  p.M = M; p.N = N; p.epi = epi; p.C = C; p.ldc = ldc; p.nseg = 1;
  return p;
}


hipError_t launch_gemm(Launch& L, bool tn, hipStream_t stream);

}  // namespace gh
