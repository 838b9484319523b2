// Fused entry points built on the grouped MFMA GEMM: GGNN cell fwd/bwd, concat attention fwd/bwd,
// plain linear fwd/bwd.  See include/get_hip.h for the contract of each.
#include "../../include/get_hip.h"
#include "common.h"
#include "gemm.hip.h"
#include "gemm_nt.hip.h"
#include "gemm_tn.hip.h"
#include "gemm_tn_pp.hip.h"
#include <stdlib.h>

namespace gh {

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static bool s_lds_bad(const Problem& p) { return p.seg[0].lda % 8 || p.seg[0].ldb % 8; }
// fast-path preconditions of gemm_nt.hip.h (NT: both operands contraction-contiguous) and gemm_tn.hip.h (TN); both LDS-DMA
static void fast_why(int rule, int prob) {      // tool build: GH_FASTOK_DEBUG=1 names the rule that sent a launch to the generic kernel
  static int dbg = -1;
  if (dbg < 0) dbg = measure_env("GH_FASTOK_DEBUG", 0);
  if (dbg) fprintf(stderr, "gemm: generic kernel (fast_ok rule %d, problem %d)\n", rule, prob);
}
static bool fast_ok(const Launch& L, bool tn) {
  for (int i = 0; i < L.nprob; ++i) {
    const Problem& p = L.p[i];
    const int ns = tn ? 1 : p.nseg;
    for (int j = 0; j < ns; ++j) {
      const Seg& s = p.seg[j];
      if (!s.vecA || !s.vecB || s.gatherB || s.K < 4) { fast_why(1, i); return false; }
      if (tn && s.gatherA) { fast_why(2, i); return false; }
    }
    if (p.elt && tn && (p.M % 8 || p.N % 8 || s_lds_bad(p))) { fast_why(3, i); return false; }
    if (!tn) {  // operands are addressed through buffer descriptors: 31-bit byte offsets
      for (int j = 0; j < p.nseg; ++j) {
        if (4.0 * (double)p.seg[j].ldb * (double)p.N >= 2147483648.0) { fast_why(4, i); return false; }
        if (!p.seg[j].gatherA && 4.0 * (double)p.seg[j].lda * (double)p.M >= 2147483648.0) { fast_why(5, i); if (measure_env("GH_FASTOK_DEBUG", 0)) fprintf(stderr, "   seg %d lda %d M %d N %d K %d\n", j, p.seg[j].lda, p.M, p.N, p.seg[j].K); return false; }   // (gathered tables: < 2 GB by contract)
      }
    }
    if (tn) {
      const Seg& s = p.seg[0];
      if (4.0 * (double)s.lda * (double)s.K >= 2147483648.0 || 4.0 * (double)s.ldb * (double)s.K >= 2147483648.0) { fast_why(6, i); return false; }
    }
    if (p.N % 4 || p.N < 4 || p.ldc % 4 || !al16(p.C)) { fast_why(7, i); return false; }
    if (tn && (p.M % 4 || p.M < 4)) { fast_why(8, i); return false; }
    if ((p.bias && !al16(p.bias)) || (p.bias2 && !al16(p.bias2)) || (p.out1 && !al16(p.out1)) || (p.in0 && !al16(p.in0)) || (p.in1 && !al16(p.in1)))
      { fast_why(9, i); return false; }
    if (p.epi == EPI_ATT && (!al16(p.u) || p.ldu % 4 || !al16(p.w2))) { fast_why(10, i); return false; }
    if (p.epi == EPI_TANH_H && p.w2 && !al16(p.w2)) { fast_why(11, i); return false; }
    if (p.epi == EPI_GATE_PRE && (tn || !p.in0 || !p.in1 || !p.in2 || !p.out1 || !p.out2 || !al16(p.in2) || !al16(p.out2) ||
                                  (p.gin && !al16(p.gin)))) { fast_why(12, i); return false; }
  }
  return true;
}
// bf16 weight-gradient GEMM on 128 x 320 tiles (gemm_tn_bf16_kernel<2, 2, 10, 4>; tool build: GH_TN16_TALL=0 restores 64 x 320)
static bool tn16_tall() {
  static int v = -1;
  if (v < 0) v = measure_env("GH_TN16_TALL", 1);
  return v != 0;
}
// bf16 weight-gradient GEMM on the 256 x 256 x 64 ping-pong tile (gemm_tn_pp.hip.h): K-chunk rows from which a batch of whole-output
// problems takes it (tool build: GH_TN_PP_ROWS; a huge value restores the 128 x 320 kernel everywhere)
static int tn_pp_rows() {
  static int v = -1;
  if (v < 0) v = measure_env("GH_TN_PP_ROWS", 16384);
  return v;
}
static bool wants_dropout(const Launch& L) {
  for (int i = 0; i < L.nprob; ++i)
    if (L.p[i].drop_mode) return true;
  return false;
}

// launches since the last reset: {fast kernel, generic kernel, generic kernel with >= 1 GFLOP of work}.  The generic
// kernel is the correctness net for odd shapes and unaligned operands; a LARGE GEMM landing on it is a performance
// bug upstream (e.g. a misaligned parameter view), which tests assert against through gh_gemm_path_counters.
static long long g_path_counts[3] = {0, 0, 0};
static int g_gemm_mode = 0;       // 0: fp32 MFMA everywhere (default); 1: bf16 MFMA in the big-tile NT/NN GEMMs; 2: fp32x3; 3: fp32x3 with pre-split weights (gh_set_gemm_mode)
// mode 3: replaces the B operands (weights) of an NT launch by their pre-split images (registry below); false = not every
// operand could be served, the launch runs the in-register split (MODE 3) instead
static bool x3p_substitute(Launch& L, hipStream_t s);

int gemm_mode() { return g_gemm_mode; }

static bool few_row_bf16() {
  static int v = -1;
  if (v < 0) v = measure_env("GH_FEW_BF16", 1);
  return v != 0;
}

template <int WM, int WN, int NI, int MI = 2>
static hipError_t launch_cfg(const Launch& L, bool tn, hipStream_t s) {
  const bool fast = fast_ok(L, tn);
  {
    double fl = 0.0;
    for (int i = 0; i < L.nprob; ++i)
      for (int j = 0; j < L.p[i].nseg; ++j) fl += 2.0 * L.p[i].M * L.p[i].N * (double)L.p[i].seg[j].K;
    g_path_counts[fast ? 0 : 1] += 1;
    if (!fast && fl >= 1e9) g_path_counts[2] += 1;
  }
  const int n_outer = tn ? L.ksplit : L.m_tiles;
  const int n_inner = tn ? L.m_tiles * L.nprob : L.nprob * L.ksplit;
  const bool nt_few = !tn && L.m_tiles < 8 && fast && !(GH_DBG_BITS(L) & 32);      // gemm_nt.hip.h: linear work decode for launches of fewer than 8 row tiles
  const int grid = nt_few ? n_outer * n_inner : 8 * ((n_outer + 7) / 8) * n_inner;
  if (grid <= 0) return hipSuccess;
  // profiler row: by the SIZE of the launch, not by the tile configuration it runs on -- the activation-sized single-problem
  // launches that the occupancy rule moves to the 32-row tile belong with the big GEMMs (booked under
  // gemm_small they made gemm_big look better than the path is)
  int max_rows = 0;
  for (int i = 0; i < L.nprob; ++i) { const int rws = tn ? L.p[i].seg[0].K : L.p[i].M; if (rws > max_rows) max_rows = rws; }
  const int tag = (max_rows >= 8192 ? PROF_GEMM_BIG : PROF_GEMM_SMALL) + (tn ? 1 : 0);
  double flops = 0.0;
  if (prof_enabled()) {
    for (int i = 0; i < L.nprob; ++i)
      for (int j = 0; j < L.p[i].nseg; ++j) {
        int rows = L.p[i].M;       // segment 0 is skipped by the row tiles at or beyond seg0_rows: do not count it
        if (j == 0 && !tn && L.p[i].nseg > 1 && L.p[i].seg0_rows > 0) {
          const int bm = 16 * MI * WM, cut = ((L.p[i].seg0_rows + bm - 1) / bm) * bm;
          if (cut < rows) rows = cut;
        }
        flops += 2.0 * rows * L.p[i].N * (double)L.p[i].seg[j].K;
      }
    prof_begin(s, tag);
  }
  if (!fast && wants_dropout(L)) return hipErrorInvalidValue;     // fused dropout exists in the fast kernels only
  if (!fast && !tn)
    for (int i = 0; i < L.nprob; ++i)
      if ((L.p[i].epi == EPI_TANH_H && L.p[i].w2) || L.p[i].epi == EPI_GATE_PRE || (L.p[i].epi == EPI_ATT && L.p[i].e_atomic == 2))
        return hipErrorInvalidValue;   // so do the fused scorer projection, the fused gate head and the head scores of column blocks
  bool launched = false;
  bool any_elt = false, all_elt = true;
  for (int i = 0; i < L.nprob; ++i) { any_elt = any_elt || L.p[i].elt; all_elt = all_elt && L.p[i].elt; }
  if (any_elt) {      // bf16 storage pipeline: only on the 64x320 fast kernels, never mixed with fp32 problems
    if (!fast || !all_elt) return hipErrorInvalidValue;
    if constexpr (WM == 2 && WN == 2 && NI == 10) {
      if (tn) {
        if (tn16_tall()) hipLaunchKernelGGL((gemm_tn_bf16_kernel<2, 2, 10, 4>), dim3(grid), dim3(256), 0, s, L);
        else hipLaunchKernelGGL((gemm_tn_bf16_kernel<2, 2, 10>), dim3(grid), dim3(256), 0, s, L);
      }
      else hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 10, 2, 2>), dim3(grid), dim3(256), 0, s, L);
      launched = true;
    } else if constexpr (WM == 2 && WN == 2 && NI == 8 && MI == 4) {      // 128 x 256 tile (NT only)
      if (tn) return hipErrorInvalidValue;
      constexpr int kLds = 3 * (128 + 256) * 64;          // three K-loop stages (the epilogue staging fits inside)
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<2, 2, 8, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds); attr = true; }
      hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 8, 4, 2>), dim3(grid), dim3(256), kLds, s, L);
      launched = true;
    } else if constexpr (WM == 2 && WN == 2 && NI == 4 && MI == 4) {      // 128 x 128 tile, three workgroups per CU (NT only)
      if (tn) return hipErrorInvalidValue;
      constexpr int kLds = 3 * (128 + 128) * 64;
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<2, 2, 4, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds); attr = true; }
      hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 4, 4, 2>), dim3(grid), dim3(256), kLds, s, L);
      launched = true;
    } else if constexpr (WM == 2 && WN == 4 && NI == 4 && MI == 8) {      // 256 x 256 x 64 tile, 8 waves, ping-pong K loop (NT only; gemm_nt_pp.hip.h)
      if (tn) return hipErrorInvalidValue;
      constexpr int kLds = 131072;                        // two K tiles of 64 KB (the epilogue staging fits inside)
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<2, 4, 4, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds); attr = true; }
      hipLaunchKernelGGL((gemm_nt_kernel<2, 4, 4, 8, 2>), dim3(grid), dim3(512), kLds, s, L);
      launched = true;
    } else return hipErrorInvalidValue;
  } else if (MI >= 4) {
    return hipErrorInvalidValue;          // the 128 x 256 / 256 x 256 tiles exist for the bf16 storage pipeline only
  }
  // (the rest of this function only exists for the MI = 2 configurations: as plain run-time branches the fp32 / fp32x3 / generic
  //  launches below were instantiated for the bf16-only tile shapes as well -- nine never-launched kernels, half of this file's build time)
  if constexpr (MI == 2) {
  if (launched) {
  } else if (fast && !tn) {
    // gh_set_gemm_mode(1): bf16 operand rounding in the activation-sized (64x320 tile) launches only
    if (g_gemm_mode == 1 && WM == 2 && WN == 2 && NI == 10) {
      if constexpr (WM == 2 && WN == 2 && NI == 10)
        hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 10, 2, true>), dim3(grid), dim3(256), 0, s, L);
    } else if (g_gemm_mode == 1 && WM == 1 && WN == 4 && NI == 5 && few_row_bf16()) {
      // ... and in the few-row launches (32 x 320 tile): at h = 768 the evidence-level products (960 rows against 22 / 41 MB
      // weights) and the claim cell are 0.5 ms of fp32 MFMA time per step
      if constexpr (WM == 1 && WN == 4 && NI == 5)
        hipLaunchKernelGGL((gemm_nt_kernel<1, 4, 5, 2, true>), dim3(grid), dim3(256), 0, s, L);
    } else if constexpr (WM == 4) {      // the ping-pong fp32 tile exists in the exact mode only (Batch sets pp32 for g_gemm_mode == 0)
      if (g_gemm_mode != 0) return hipErrorInvalidValue;
      hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, NI, 2>), dim3(grid), dim3(WM * WN * 64), 0, s, L);
    } else if (g_gemm_mode == 2 || g_gemm_mode == 3) {      // fp32x3 (experimental): fp32 values and results, products from 3-way bf16 splits on the bf16 MFMA
      Launch L2;
      bool pre = false;
      if (g_gemm_mode == 3) { L2 = L; pre = x3p_substitute(L2, s); }
      if (pre) {
        constexpr int kLds = 2 * (16 * 2 * WM * 64 + 16 * NI * WN * 112);      // two K-loop stages, B rows at the 112-byte pitch
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<WM, WN, NI, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds); attr = true; }
        hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, NI, 2, 4>), dim3(grid), dim3(WM * WN * 64), kLds, s, L2);
      } else
        hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, NI, 2, 3>), dim3(grid), dim3(WM * WN * 64), 0, s, L);
    } else
      hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, NI, 2>), dim3(grid), dim3(WM * WN * 64), 0, s, L);
    launched = true;
  } else if (fast && tn) {
    if constexpr (WM == 2 && WN == 2 && NI == 10) {        // weight gradients always take the 64x320 tile (Batch: big = tn || ...)
      if (g_gemm_mode == 1) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 10, true>), dim3(grid), dim3(256), 0, s, L);
      else hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 10>), dim3(grid), dim3(256), 0, s, L);
      launched = true;
    }
  }
  if (!launched) {
    if (tn) hipLaunchKernelGGL((gemm_kernel<WM, WN, NI, true>), dim3(grid), dim3(WM * WN * 64), 0, s, L);
    else hipLaunchKernelGGL((gemm_kernel<WM, WN, NI, false>), dim3(grid), dim3(WM * WN * 64), 0, s, L);
  }
  }      // MI == 2
  if (!launched && MI != 2) return hipErrorInvalidValue;
  prof_end(tag, flops, s);
  return hipGetLastError();
}

// bf16 weight gradients on the 256 x 256 x 64 ping-pong tile: one workgroup per CU, work items dealt to the XCDs in runs (L.per each)
static hipError_t launch_tn_pp(const Launch& L, hipStream_t s) {
  const int grid = 8 * L.per;
  if (grid <= 0) return hipSuccess;
  g_path_counts[0] += 1;
  const int tag = PROF_GEMM_BIG + 1;
  double flops = 0.0;
  if (prof_enabled()) {
    for (int i = 0; i < L.nprob; ++i) flops += 2.0 * L.p[i].M * L.p[i].N * (double)L.p[i].seg[0].K;
    prof_begin(s, tag);
  }
  constexpr int kLds = 131072;      // two K tiles of 64 KB
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_tn_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLds); attr = true; }
  hipLaunchKernelGGL(gemm_tn_pp_kernel, dim3(grid), dim3(512), kLds, s, L);
  prof_end(tag, flops, s);
  return hipGetLastError();
}

// Split-K scratch registered by the caller (gh_set_workspace); partial tiles are written with plain
// stores and summed by reduce_partials_kernel instead of cross-XCD fp32 atomics.
// Registry: one default buffer plus any number of per-stream buffers (two streams that both split K must not share
// one scratch area); looked up by the launch's stream on every call.
}  // namespace gh
#include <mutex>
#include <unordered_map>
#include <vector>
namespace gh {
// ---- fp32x3 with pre-split weights (gh_set_gemm_mode(3)) ---------------------------------------------
// Image of a weight view B[N][ldb] (K columns used): per row ceil(K / 16) K tiles of 96 bytes, each four groups of
// {hi[4], mid[4], lo[4]} bf16 -- the three pieces of four consecutive k (zeros beyond K).  gemm_nt_kernel MODE 4 DMAs a
// K tile of a row as six 16-byte chunks.  Images are made on first use (on the launching stream), re-made in one batched
// launch by gh_fp32x3_refresh after the optimiser step, and individually when found stale (gh_weights_changed without
// a refresh: correct, slow).
struct X3Item { const float* src; unsigned char* dst; int N, ldb, K, pitch; };
constexpr int X3_BATCH = 64;
struct X3Args { int n; X3Item it[X3_BATCH]; };
__global__ void __launch_bounds__(256) split3_kernel(const X3Args a) {
  const X3Item& it = a.it[blockIdx.y];
  const int G = it.pitch / 24;
  const long long total = (long long)it.N * G;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int row = (int)(idx / G), g = (int)(idx - (long long)row * G);
    const float* src = it.src + (size_t)row * it.ldb + 4 * g;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (4 * g + e < it.K) ? src[e] : 0.f;
    unsigned hi[2], mid[2], lo[2];
    hi[0] = nt_pack_bf16(v[0], v[1]); hi[1] = nt_pack_bf16(v[2], v[3]);
    const float r0 = v[0] - __builtin_bit_cast(float, hi[0] << 16), r1 = v[1] - __builtin_bit_cast(float, hi[0] & 0xffff0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, hi[1] << 16), r3 = v[3] - __builtin_bit_cast(float, hi[1] & 0xffff0000u);
    mid[0] = nt_pack_bf16(r0, r1); mid[1] = nt_pack_bf16(r2, r3);
    const float q0 = r0 - __builtin_bit_cast(float, mid[0] << 16), q1 = r1 - __builtin_bit_cast(float, mid[0] & 0xffff0000u);
    const float q2 = r2 - __builtin_bit_cast(float, mid[1] << 16), q3 = r3 - __builtin_bit_cast(float, mid[1] & 0xffff0000u);
    lo[0] = nt_pack_bf16(q0, q1); lo[1] = nt_pack_bf16(q2, q3);
    uint2* dst = reinterpret_cast<uint2*>(it.dst + (size_t)row * it.pitch + (size_t)g * 24);
    dst[0] = make_uint2(hi[0], hi[1]); dst[1] = make_uint2(mid[0], mid[1]); dst[2] = make_uint2(lo[0], lo[1]);
  }
}
struct X3Key {
  const void* b; int ldb, K, N;
  bool operator==(const X3Key& o) const { return b == o.b && ldb == o.ldb && K == o.K && N == o.N; }
};
struct X3Hash {
  size_t operator()(const X3Key& k) const {
    size_t h = std::hash<const void*>()(k.b);
    h ^= std::hash<long long>()(((long long)k.ldb << 40) ^ ((long long)k.K << 20) ^ (long long)k.N) + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    return h;
  }
};
struct X3Ent { unsigned char* img; int pitch; long long epoch; };
static std::mutex g_x3_mu;
static std::unordered_map<X3Key, X3Ent, X3Hash> g_x3;
static std::unordered_multimap<size_t, unsigned char*> g_x3_pool;      // buffers of cleared images, by size (hipMalloc / hipFree are slow)
static long long g_x3_epoch = 0;

static void x3p_split(const X3Item* items, int n, hipStream_t s) {
  for (int i0 = 0; i0 < n; i0 += X3_BATCH) {
    X3Args a;
    a.n = n - i0 < X3_BATCH ? n - i0 : X3_BATCH;
    long long mx = 0;
    for (int i = 0; i < a.n; ++i) {
      a.it[i] = items[i0 + i];
      const long long t = (long long)a.it[i].N * (a.it[i].pitch / 24);
      if (t > mx) mx = t;
    }
    int gx = (int)((mx + 255) / 256);
    if (gx > 256) gx = 256;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(split3_kernel, dim3(gx, a.n), dim3(256), 0, s, a);
  }
}

static bool x3p_substitute(Launch& L, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_x3_mu);
  X3Item todo[2 * GH_MAX_PROBLEMS];
  int nt = 0;
  for (int i = 0; i < L.nprob; ++i) {
    Problem& q = L.p[i];
    if (q.elt) return false;
    for (int j = 0; j < q.nseg; ++j) {
      Seg& g = q.seg[j];
      if (g.K <= 0 || g.gatherB) return false;
      const X3Key key{g.B, g.ldb, g.K, q.N};
      auto f = g_x3.find(key);
      if (f == g_x3.end()) {
        if (g_x3.size() >= 512) return false;      // views keep changing address (caller never clears): stop growing, in-register split instead
        X3Ent e;
        e.pitch = ((g.K + 15) / 16) * 96;
        e.epoch = -1;
        const size_t bytes = (size_t)q.N * e.pitch + 256;
        auto pf = g_x3_pool.find(bytes);
        if (pf != g_x3_pool.end()) { e.img = pf->second; g_x3_pool.erase(pf); }
        else if (hipMalloc((void**)&e.img, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        f = g_x3.emplace(key, e).first;
      }
      if (f->second.epoch != g_x3_epoch) {
        todo[nt++] = X3Item{g.B, f->second.img, q.N, g.ldb, g.K, f->second.pitch};
        f->second.epoch = g_x3_epoch;
      }
      g.B = reinterpret_cast<const float*>(f->second.img);
      g.ldb = f->second.pitch / 4;
    }
  }
  if (nt > 0) x3p_split(todo, nt, s);
  return true;
}
static std::mutex g_ws_mu;
static Workspace g_ws_default = {nullptr, 0, nullptr, 0};
// keyed by (device, stream): the default stream has handle 0 on EVERY device, so the stream alone does not identify a
// scratch area in a process that drives several devices
struct WsKey { int dev; hipStream_t s; bool operator==(const WsKey& o) const { return dev == o.dev && s == o.s; } };
struct WsKeyHash { size_t operator()(const WsKey& k) const { return std::hash<const void*>()((const void*)k.s) * 31u + (size_t)(k.dev + 1); } };
static std::unordered_map<WsKey, Workspace, WsKeyHash> g_ws_stream;
static int current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) d = 0; return d; }
static Workspace make_workspace(void* ptr, int64_t bytes) {
  // the last 1/16 of the buffer (16-byte aligned) serves the column-sum partials, the rest split-K tiles
  Workspace w = {nullptr, 0, nullptr, 0};
  if (!ptr || bytes < (1 << 20)) return w;
  const size_t tail = ((size_t)bytes / 16) & ~(size_t)15;
  w.p = (float*)ptr;
  w.bytes = ((size_t)bytes - tail) & ~(size_t)15;
  w.cs = (float*)((char*)ptr + w.bytes);
  w.cs_bytes = tail;
  return w;
}
Workspace workspace_for(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  auto it = g_ws_stream.find(WsKey{current_device(), s});
  return it != g_ws_stream.end() ? it->second : g_ws_default;
}

struct ReduceItem { const float* ws; float* out; int I, J, ldc, ks; long long stride; };
struct ReduceArgs { ReduceItem it[3 * GH_MAX_PROBLEMS]; int n; };     // tiles of <= 8 problems + up to two bias-gradient outputs each

__global__ void __launch_bounds__(256)
reduce_partials_kernel(const ReduceArgs R) {
  const ReduceItem& it = R.it[blockIdx.y];
  const int J4 = it.J / 4;
  const size_t total = (size_t)it.I * J4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / J4), j = 4 * (int)(e % J4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p = it.ws + (size_t)i * it.J + j;
    int k = 0;
    for (; k + 8 <= it.ks; k += 8) {           // 8 independent 16-byte loads in flight, fixed summation order
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(k + u) * it.stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; k < it.ks; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * it.stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float* o = it.out + (size_t)i * it.ldc + j;     // += : gradients accumulate into the caller's buffer
    o[0] += acc.x; o[1] += acc.y; o[2] += acc.z; o[3] += acc.w;
  }
}

// Same reduction for FEW output elements and MANY partials (dW2 of the attention: 5 x 300 values, one partial per
// pair): one wave per float4 element, lanes stride over the partials, fixed-order shuffle tree.
__global__ void __launch_bounds__(256)
reduce_partials_wave_kernel(const ReduceArgs R) {
  const ReduceItem& it = R.it[blockIdx.y];
  const int J4 = it.J / 4;
  const int total = it.I * J4;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= total) return;
  const int i = e / J4, j = 4 * (e % J4);
  const float* p = it.ws + (size_t)i * it.J + j;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = lane; k < it.ks; k += 64) {
    const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * it.stride);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  for (int o = 32; o > 0; o >>= 1) {
    acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
  }
  if (lane == 0) {
    float* o = it.out + (size_t)i * it.ldc + j;
    o[0] += acc.x; o[1] += acc.y; o[2] += acc.z; o[3] += acc.w;
  }
}

// Second half of an NT GEMM that was split over K: sum the partial tiles, then apply the problem's real epilogue
// (any of gemm.hip.h's row epilogues).  One wave per output row.
struct FinishItem {
  const float* ws; long long stride; int ks; int M, N, epi, accumulate, ldc;
  float* C; const float* bias; const float* bias2; float* out1; const float* in0; const float* in1;
  const float* u; const float* w2; float* e; int ldu, R, heads;
  const int32_t* rowg;
};

// Two shapes.  split = 1 (few-row products with many partials: the head's K = 3556 product has 55): one WORKGROUP per output
// row, the four waves each sum a quarter of the partial tiles (eight 16-byte loads in flight per lane), the quarter sums meet
// in LDS in wave order (fixed summation order: deterministic) and wave 0 applies the epilogue -- one wave per row walked all
// `ks` partials alone, 14 dependent memory round trips = 18 us for the head; four waves need 4.  split = 0: one wave per
// row, four rows per workgroup (few partials, many rows).
struct FinishArgs { FinishItem it[GH_MAX_PROBLEMS]; int n; int split; };

__global__ void __launch_bounds__(256)
nt_finish_kernel(const FinishArgs F) {
  const FinishItem& it = F.it[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool split = F.split != 0;
  const int row = split ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
  if (row >= it.M) return;           // (workgroup-uniform when split)
  const int N4 = it.N / 4;
  __shared__ float4 part[3][64];
  int kb = 0, ke = it.ks;
  if (split) { const int kq = (it.ks + 3) / 4; kb = wave * kq; ke = min(it.ks, kb + kq); }
  float pe[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) pe[c] = 0.f;
  for (int c0 = 0; c0 < N4; c0 += 64) {
    const int c4 = c0 + lane;
    const int col = 4 * c4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < N4 && kb < ke) {
      const float* p = it.ws + (size_t)row * it.N + col;
      for (int k0 = kb; k0 < ke; k0 += 8) {            // eight partial tiles in flight, added in split order
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(p + (size_t)min(k0 + u, ke - 1) * it.stride);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < ke) { v.x += x[u].x; v.y += x[u].y; v.z += x[u].z; v.w += x[u].w; }
      }
    }
    if (split) {
      if (c0 > 0) __syncthreads();                     // wave 0 is done with the previous chunk's quarter sums
      if (wave > 0) part[wave - 1][lane] = v;
      __syncthreads();
      if (wave > 0) continue;
#pragma unroll
      for (int w = 0; w < 3; ++w) { const float4 q = part[w][lane]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    }
    if (c4 >= N4) continue;
    const size_t oo = (size_t)row * it.ldc + col;
    float* o = it.C + oo;
    if (it.bias) { const float4 b4 = *reinterpret_cast<const float4*>(it.bias + col); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
    if (it.bias2) { const float4 b4 = *reinterpret_cast<const float4*>(it.bias2 + col); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
    if (it.epi == EPI_ATT) {
      const float4 u4 = *reinterpret_cast<const float4*>(it.u + (size_t)(it.rowg ? it.rowg[row] : row / it.R) * it.ldu + col);
      const float4 t = make_float4(tanhf_(v.x + u4.x), tanhf_(v.y + u4.y), tanhf_(v.z + u4.z), tanhf_(v.w + u4.w));
      *reinterpret_cast<float4*>(o) = t;
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < it.heads) {
          const float4 w = *reinterpret_cast<const float4*>(it.w2 + (size_t)c * it.N + col);
          pe[c] += t.x * w.x + t.y * w.y + t.z * w.z + t.w * w.w;
        }
    } else if (it.epi == EPI_SIGMOID_Z) {
      *reinterpret_cast<float4*>(o) = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
    } else if (it.epi == EPI_SIGMOID_R) {
      const float4 x = *reinterpret_cast<const float4*>(it.in0 + oo);
      const float4 r4 = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
      *reinterpret_cast<float4*>(o) = r4;
      *reinterpret_cast<float4*>(it.out1 + oo) = make_float4(r4.x * x.x, r4.y * x.y, r4.z * x.z, r4.w * x.w);
    } else if (it.epi == EPI_TANH_H) {
      const float4 z = *reinterpret_cast<const float4*>(it.in0 + oo);
      const float4 x = *reinterpret_cast<const float4*>(it.in1 + oo);
      const float4 h = make_float4(tanhf_(v.x), tanhf_(v.y), tanhf_(v.z), tanhf_(v.w));
      *reinterpret_cast<float4*>(o) = h;
      *reinterpret_cast<float4*>(it.out1 + oo) = make_float4(h.x * z.x + x.x * (1.f - z.x), h.y * z.y + x.y * (1.f - z.y),
                                                             h.z * z.z + x.z * (1.f - z.z), h.w * z.w + x.w * (1.f - z.w));
    } else if (it.epi == EPI_BWD_DRX) {
      const float4 x = *reinterpret_cast<const float4*>(it.in0 + oo);
      const float4 r4 = *reinterpret_cast<const float4*>(it.in1 + oo);
      float4 d = *reinterpret_cast<const float4*>(it.out1 + oo);
      *reinterpret_cast<float4*>(o) = make_float4(v.x * x.x * r4.x * (1.f - r4.x), v.y * x.y * r4.y * (1.f - r4.y),
                                                  v.z * x.z * r4.z * (1.f - r4.z), v.w * x.w * r4.w * (1.f - r4.w));
      d.x += v.x * r4.x; d.y += v.y * r4.y; d.z += v.z * r4.z; d.w += v.w * r4.w;
      *reinterpret_cast<float4*>(it.out1 + oo) = d;
    } else {
      if (it.accumulate) {
        const float4 c4v = *reinterpret_cast<const float4*>(o);
        v.x += c4v.x; v.y += c4v.y; v.z += c4v.z; v.w += c4v.w;
      }
      *reinterpret_cast<float4*>(o) = v;
      if (it.w2) {      // a second, few-column linear layer on the finished row (the head's out[1]): e = v . w2^T (+ in0 as its bias)
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < it.heads) {
            const float4 w = *reinterpret_cast<const float4*>(it.w2 + (size_t)c * it.N + col);
            pe[c] += v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
          }
      }
    }
  }
  const bool proj = it.epi == EPI_STORE && it.w2 != nullptr;
  if ((it.epi == EPI_ATT || proj) && (!split || wave == 0)) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float v = pe[c];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0 && c < it.heads) it.e[(size_t)row * it.heads + c] = v + ((proj && it.in0) ? it.in0[c] : 0.f);
    }
  }
}

// Collects problems that share their row space, splits wide outputs into column blocks the tile
// can hold, and launches them GH_MAX_PROBLEMS at a time.
constexpr int NARROW_DEFAULT = 14;     // call-site mask of the 64 x 160 tile (see Batch::narrow): cell forward h gate (2), cell backward da / d(r xp) pair (3) and da / dxp accumulation pair (4).  A/B on the bench step: 153.3 K -> 155.2 K pairs/s; the single-problem K = 300 sites (5, 6, 7) and the z / r pair (1) measured -0.1 .. +0.1 %
struct Batch {
  Launch L;
  bool tn, big;
  hipStream_t s;
  int bm, bn;
  hipError_t err = hipSuccess;
  int k_total = 0;
  // TN: bias gradients fused into the weight-gradient launch (Problem::colsum): per problem up to two outputs that
  // receive (+=) the column sums of the A operand.  colsum_fused tells the caller whether that happened.
  float* cs_out[GH_MAX_PROBLEMS] = {nullptr};
  float* cs_out2[GH_MAX_PROBLEMS] = {nullptr};
  bool colsum_fused = false;
  bool colsum_missed = false;   // a requested column sum could be produced neither by the launch nor by the fallback kernel (bf16 operands without workspace)
  bool split_done = false;      // the last flush took the few-row split-K plan (its finish kernel applied the problems' row epilogues)
  void want_colsum(float* o, float* o2) { if (L.nprob > 0) { cs_out[L.nprob - 1] = o; cs_out2[L.nprob - 1] = o2; } }

  float* g_ws = nullptr;
  size_t g_ws_bytes = 0;
  // 64 x 160 fp32 tile (launch_cfg<2, 2, 5>): half the accumulators -> <= 128 VGPRs -> FOUR workgroups per CU instead of three,
  // two column blocks per 300-wide problem.  More resident workgroups run more of the neighbours' epilogue streams underneath
  // the K loops: launches with stream-heavy epilogues gain, plain-store launches lose (prototype, tools/glds_proto.hip NHALF:
  // h-gate-like epilogue K = 300: 62.6 -> 68.2 TF, K = 600: 87.7 -> 90.5; plain store: 92.4 -> 88.7).  Chosen per call site.
  bool narrow = false;
  // 128 x 160 fp32 tile, 8 waves, two workgroups per CU, ping-pong K loop (launch_cfg<4, 2, 5>; gemm_nt_pp32.hip.h).  TOOL BUILD ONLY
  // (GH_PP32_ROWS): built in round 6 as the fp32 half of VERDICT r5 item 1 and measured SLOWER than the 64 x 320 / 64 x 160 tiles at three
  // / four workgroups per CU: K loops alone 3.37 against 3.15 ms per step, whole launches 3.98 against 3.72 (two and three LDS buffers
  // alike; a half-K-loop start stagger of each CU's second workgroup: no effect at any amount).  DESIGN 4.5.
  bool pp32 = false;
  // EPI_ATT on rows wider than one column block (h = 768): every block applies tanh(. + u) to its columns and reduces ITS share
  // of the head scores W2 . t into a partial buffer of its own, e + block * e_block_stride (plain stores: the consumer,
  // att_softmax_fwd, adds the partials in block order -- deterministic).  0 = whole rows only.
  long long e_block_stride = 0;
  bool wide = false;      // 128 x 256 bf16 tile (launch_cfg<2, 2, 8, 4>)
  bool wide256 = false;   // 256 x 256 x 64 bf16 tile, 8 waves, ping-pong K loop (launch_cfg<2, 4, 4, 8>)
  // weight gradients of the bf16 storage pipeline on the 256 x 256 x 64 ping-pong tile (gemm_tn_pp.hip.h): the batch's problems stay
  // WHOLE outputs (no 320-column blocks), the kernel walks their 256 x 256 tiles.  Decided by the first problem of a launch.
  bool tn_pp = false;
  bool tn_pp_ok(const Problem& p) const {
    if (!tn || !p.elt || p.nseg != 1 || g_ws == nullptr || p.epi != EPI_ATOMIC) return false;
    const Seg& sg = p.seg[0];
    if (sg.gatherA || sg.gatherB || !sg.vecA || !sg.vecB || sg.K < tn_pp_rows()) return false;
    if (p.M % 256 || p.N % 256 || p.ldc % 4 || (reinterpret_cast<uintptr_t>(p.C) & 15)) return false;
    return (size_t)sg.K * (size_t)sg.lda * 2 < ((size_t)1 << 31) && (size_t)sg.K * (size_t)sg.ldb * 2 < ((size_t)1 << 31);
  }
  bool wide128 = false;   // 128 x 128 bf16 tile, three workgroups per CU (launch_cfg<2, 2, 4, 4>; tool build: GH_BF16_TILE=128)
  // (round 5: the same 128 x 256 tile on EIGHT waves -- 64 x 64 per wave, two workgroups = four waves per SIMD -- needs 146 VGPRs
  //  for its 128-register budget: 82 spills, 16 instead of 12 ds_read_b128 per 32 MFMAs; configs[4] bf16 115.3 -> 99.6 K pairs/s. Removed.)
  // wide_bf16: every problem of this batch is a bf16-storage NT problem whose widths are multiples of 256 (h = 768)
  // site: 0 = never narrow; 1.. = call site id, narrow when the site's bit is set in the mask (tool build: GH_NT_NARROW)
  // n_hint: output width of the site's problems -- widths that 160-column blocks cover with less padding than 320-column
  // blocks (h = 768: 800 against 960 computed columns) take the narrow tile at every site (configs[4] in fp32: 26.1 -> 28.4 K pairs/s)
  Batch(bool tn_, int rows_hint, hipStream_t s_, bool wide_bf16 = false, int site = 0, int n_hint = 0) : tn(tn_), s(s_) {
    const Workspace w = workspace_for(s_);
    g_ws = w.p; g_ws_bytes = w.bytes;
    static int force_small = -1;
    if (force_small < 0) force_small = measure_env("GH_NT_FORCE_SMALL", 0);
    big = tn_ || (rows_hint >= 8192 && !force_small);      // 64x320 tile (2x2 waves) for the activation-sized GEMMs, 32x320 (1x4) for few-row ones
    bm = big ? 64 : 32;
    bn = 320;
    static int no_wide = -1;
    if (no_wide < 0) no_wide = measure_env("GH_BF16_TILE", 0) == 320 ? 1 : 0;
    if (wide_bf16 && big && !tn_ && !no_wide) { wide = true; bm = 128; bn = 256; }
    // 256 x 256 x 64 / 8 waves, one workgroup per CU, ping-pong K loop + software-pipelined epilogue (gemm_nt_pp.hip.h, round 6):
    // the default for activation-sized launches of the bf16 storage pipeline -- configs[4], B = 32: gemm_big 4.25 -> 3.09 ms per
    // step, 128.1 -> 148.6 K pairs/s.  (Round 3's 256 x 256 tile -- the 128 x 256 tile's K loop on 8 waves, 32-deep stages --
    // measured 0.200 against 0.207: not the tile size but the loop structure is the lever.)  Launches below 32 768 rows (fewer
    // than 1.5 workgroups per CU) keep the 128 x 256 tile.  Tool build: GH_BF16_TILE=1 restores 128 x 256 everywhere,
    // GH_BF16_TILE256_ROWS moves the row threshold.
    static int tile_env = -1;
    if (tile_env < 0) tile_env = measure_env("GH_BF16_TILE", 0);
    static int tile256_rows = -1;
    if (tile256_rows < 0) tile256_rows = measure_env("GH_BF16_TILE256_ROWS", 32768);
    if (wide && rows_hint >= tile256_rows && tile_env != 1 && tile_env != 128) { wide256 = true; bm = 256; }
    static int tile128 = -1;
    if (tile128 < 0) tile128 = measure_env("GH_BF16_TILE", 0) == 128 ? 1 : 0;
    if (wide && tile128) { wide128 = true; bn = 128; }
    static int narrow_mask = -1;
    if (narrow_mask < 0) narrow_mask = measure_env("GH_NT_NARROW", NARROW_DEFAULT);
    const bool less_pad = n_hint > 0 && (n_hint + 159) / 160 * 160 < (n_hint + 319) / 320 * 320;
    // launches whose 64 x 320 grid would not fill one round of 768 workgroup slots (realistic evidence counts: 14 208 rows = 222
    // row tiles) get twice the workgroups on 1024 slots: 99.0 -> 102.0 K pairs/s on the Snopes-histogram step at B = 32
    const bool thin = rows_hint <= 24576;
    if (site > 0 && big && !tn_ && !wide && g_gemm_mode == 0 && narrow_mask != 0 && (((narrow_mask >> (site - 1)) & 1) || less_pad || thin)) { narrow = true; bn = 160; }
    static int pp32_rows = -1;
    if (pp32_rows < 0) pp32_rows = measure_env("GH_PP32_ROWS", 1 << 30);      // (tool build: row threshold of the fp32 ping-pong tile)
    // (`narrow` stays set: the call sites' column-block logic -- two 160-wide blocks per 300-wide problem, the scorer's two partial dot
    //  products -- is the 64 x 160 tile's; launches without a call-site id, e.g. the attention's t product with its head scores, keep the 64 x 320 tile)
#ifdef GH_MEASURE
    if (site > 0 && big && !tn_ && !wide && g_gemm_mode == 0 && rows_hint >= pp32_rows) { pp32 = true; narrow = true; bm = 128; bn = 160; }
#endif
    // (round 5: thin launches on 32 x 160 tiles -- one 16-row MFMA tile per wave, 80 - 96 VGPRs, five or six workgroups per CU, twice
    //  the waves for grids that fill less than one round -- measured on the Snopes-count step: 109.4 K -> 108.2 - 109.4 K pairs/s
    //  at every site mask.  Removed.)
    reset();
  }
  void reset() {
    L.nprob = 0; L.m_tiles = 0; L.ksplit = 1; L.kchunk = 0; k_total = 0; L.n_tiles = 0; L.per = 0;
    static int dbg = -1;
    if (dbg < 0) dbg = measure_env("GH_DBG", 0);
    L.dbg = dbg;
    for (int i = 0; i < GH_MAX_PROBLEMS; ++i) cs_out[i] = cs_out2[i] = nullptr;
  }

  void add(const Problem& p) {
    // row reductions need whole rows -- except the scorer's single dot product, which two column blocks may add up (e_atomic)
    const bool scorer_blocks = p.epi == EPI_TANH_H && p.w2 && p.N > bn && narrow && p.N <= 2 * bn;
    const bool att_blocks = p.epi == EPI_ATT && p.N > bn && e_block_stride > 0;
    // ... or, with a partial buffer per block (e_block_stride), any number of them: the scorer kernel adds the partials in block order
    const bool scorer_parts = p.epi == EPI_TANH_H && p.w2 && p.N > bn && !scorer_blocks && e_block_stride > 0;
    if ((p.epi == EPI_ATT || (p.epi == EPI_TANH_H && p.w2)) && p.N > bn && !scorer_blocks && !att_blocks && !scorer_parts) { err = hipErrorInvalidValue; return; }
    if (tn) {
      const bool want = tn_pp_ok(p);
      if (L.nprob > 0 && want != tn_pp) flush();      // (a launch is one kernel: problems of the other kind start the next one)
      if (L.nprob == 0) tn_pp = want;
    }
    const int bn = tn_pp ? (1 << 30) : this->bn;
    for (int n0 = 0; n0 < p.N; n0 += bn) {
      Problem q = p;
      q.N = (p.N - n0 < bn) ? p.N - n0 : bn;
      auto adv = [](const float* ptr, size_t elems, bool bf) { return (const float*)((const char*)ptr + elems * (bf ? 2 : 4)); };
      for (int i = 0; i < q.nseg; ++i) {
        if (tn) {
          q.seg[i].B = adv(q.seg[i].B, (size_t)n0, q.elt);
          q.seg[i].vecB = q.elt ? (((reinterpret_cast<uintptr_t>(q.seg[i].B) & 15) == 0) && q.seg[i].ldb % 8 == 0 && q.N % 8 == 0)
                                : vec_ok(q.seg[i].B, q.seg[i].ldb, q.N);
        } else {
          q.seg[i].B = adv(q.seg[i].B, (size_t)n0 * q.seg[i].ldb, q.elt);       // B is [N][ldb]: a column block of C is a row block of B
        }
      }
      q.C = (float*)adv(q.C, (size_t)n0, q.io & 1);
      q.drop_col0 = p.drop_col0 + n0;
      if (q.bias) q.bias += n0;
      if (q.bias2) q.bias2 += n0;
      if (q.out1) q.out1 = (float*)adv(q.out1, (size_t)n0, q.io & 2);
      if (q.in0) q.in0 = adv(q.in0, (size_t)n0, q.io & 4);
      if (q.in1) q.in1 = adv(q.in1, (size_t)n0, q.io & 8);
      if (q.c32) q.c32 += n0;
      if (q.gin) q.gin += n0;
      if (q.in2) q.in2 = adv(q.in2, (size_t)n0, q.io & 16);
      if (q.out2) q.out2 = (float*)adv(q.out2, (size_t)n0, q.io & 32);
      if (scorer_blocks) { q.w2 += n0; q.e_atomic = 1; }
      if (scorer_parts) { q.w2 += n0; q.e = p.e + (size_t)(n0 / bn) * (size_t)e_block_stride; q.e_atomic = 2; }
      if (att_blocks) { q.w2 += n0; q.u += n0; q.e = p.e + (size_t)(n0 / bn) * (size_t)e_block_stride; q.e_atomic = 2; }      // (2: own partial buffer, plain stores)
      if (L.nprob == GH_MAX_PROBLEMS) flush();
      L.p[L.nprob++] = q;
      const int bm_eff = tn_pp ? 256 : (tn && q.elt && tn16_tall()) ? 128 : bm;      // (all problems of a TN launch share the storage mode)
      const int mt = (q.M + bm_eff - 1) / bm_eff;
      if (mt > L.m_tiles) L.m_tiles = mt;
      if (tn_pp && (q.N + 255) / 256 > L.n_tiles) L.n_tiles = (q.N + 255) / 256;
      if (q.seg[0].K > k_total) k_total = q.seg[0].K;
    }
  }

  void flush() {
    if (L.nprob == 0 || err != hipSuccess) { reset(); return; }
    if (tn) {
      const int n_inner = L.m_tiles * L.nprob;
      static int target = -1;
      if (target < 0) target = measure_env("GH_TN_SPLIT_TARGET", 2304);
      int ks = (target + n_inner - 1) / n_inner;   // three resident rounds of 256 CUs x 3 workgroups (measured on the bench step: 1152 -> 2.06 ms,
                                                   // 1536 -> 1.95, 2304 -> 1.86, 3072 -> 1.84 but more partials to reduce; one round 20 % slower)
      // shortest K chunk: 256 rows (fp32).  bf16 storage, wide outputs (h = 768: 64 x 320 partial tiles of 80 KB each): a single
      // 768 x 768 product split into 64 chunks writes and re-reads more partial-tile bytes than it streams operands
      static int min_rows16 = -1;
      if (min_rows16 < 0) min_rows16 = measure_env("GH_TN_MIN_ROWS16", 4096);      // (A/B on configs[4]: 256 / 1024 / 2048 / 4096 rows = 106.76 / 106.72 / 106.81 / 107.17 K pairs/s)
      const int min_rows = L.p[0].elt ? min_rows16 : 256;
      const int ks_max = (k_total / min_rows > 1) ? k_total / min_rows : 1;
      if (ks > ks_max) ks = ks_max;
      if (ks < 1) ks = 1;
      // The K chunks are dealt round-robin to the 8 XCDs (chunk c runs on XCD c % 8, gemm_tn.hip.h): a chunk count that is
      // not a multiple of 8 leaves some XCDs one chunk (1 / 8 of their work at 65 chunks) more than the others.  Measured on
      // the bench step (A/B, one box): 65 chunks 158.6 K pairs/s, 72 chunks 159.9 K, 85 chunks 156.4 K.
      static int xcd_rule = -1;
      if (xcd_rule < 0) xcd_rule = measure_env("GH_TN_XCD_RULE", 1);
      const int align = tn_pp ? 64 : L.p[0].elt ? 32 : 16;      // K tile of the kernel (bf16 storage: 32 rows, ping-pong tile: 64)
      if (tn_pp) {
        // one workgroup per CU: as many K chunks as fill ONE round of 256 workgroups when that occupies >= 85 % of them (h = 768: a
        // cell's 7 outputs = 63 tiles x 4 chunks = 252), else two rounds; chunks of at least 1024 rows.  The XCD rule does not apply:
        // the kernel deals its items to the XCDs in runs.
        const int tiles = L.nprob * L.m_tiles * L.n_tiles;
        const int ks1 = 256 / tiles > 0 ? 256 / tiles : 1, ks2 = 512 / tiles > 0 ? 512 / tiles : 1;
        const double e1 = tiles * ks1 / 256.0, e2 = tiles * ks2 / 512.0;
        static int force = -1;
        if (force < 0) force = measure_env("GH_TN_PP_KS", 0);
        ks = (e1 >= 0.85 || e1 >= e2) ? ks1 : ks2;
        if (force > 0) ks = force;
        const int kmax = k_total / 1024 > 1 ? k_total / 1024 : 1;
        if (ks > kmax) ks = kmax;
        // (the partial tiles must fit the workspace: fewer chunks otherwise)
        size_t out_bytes = 0;
        for (int i = 0; i < L.nprob; ++i) out_bytes += ((size_t)L.p[i].M * L.p[i].N + (cs_out[i] ? (size_t)L.p[i].M : 0)) * sizeof(float);
        while (ks > 1 && (size_t)ks * out_bytes > g_ws_bytes) --ks;
      } else
      if (xcd_rule && ks >= 12) ks = ((ks + 4) / 8) * 8;
      int chunk = (k_total + ks - 1) / ks;
      chunk = ((chunk + align - 1) / align) * align;
      L.kchunk = chunk;
      L.ksplit = (k_total + chunk - 1) / chunk;
      if (tn_pp) L.per = (L.nprob * L.m_tiles * L.n_tiles * L.ksplit + 7) / 8;
      if (!tn_pp && xcd_rule && L.ksplit >= 12 && L.ksplit % 8 != 0) {      // rounding the chunk up dropped a chunk or two: stretch the chunks to the multiple of 8 below
        const int k8 = (L.ksplit / 8) * 8;
        chunk = (((k_total + k8 - 1) / k8 + align - 1) / align) * align;
        if ((k_total + chunk - 1) / chunk % 8 == 0) { L.kchunk = chunk; L.ksplit = (k_total + chunk - 1) / chunk; }
      }
      // partial tiles -> workspace when it is big enough and every output is float4-shaped
      size_t need = 0;
      bool want_cs0 = false;
      for (int i = 0; i < L.nprob; ++i) want_cs0 = want_cs0 || cs_out[i] != nullptr;
      // (a single chunk also goes through the workspace when the ping-pong kernel runs it -- it only writes partial tiles -- or when
      //  bf16 operands want their bias gradient: that rides in the fused kernels' workspace path only)
      bool ws_ok = g_ws != nullptr && (L.ksplit > 1 || tn_pp || (want_cs0 && L.p[0].elt));
      bool any_cs = false;
      for (int i = 0; i < L.nprob && ws_ok; ++i) {
        if (L.p[i].N % 4) ws_ok = false;
        need += (size_t)L.ksplit * L.p[i].M * L.p[i].N * sizeof(float);
        if (cs_out[i]) { need += (size_t)L.ksplit * L.p[i].M * sizeof(float); any_cs = true; }
      }
      bool want_cs = false;
      for (int i = 0; i < L.nprob; ++i) want_cs = want_cs || cs_out[i] != nullptr;
      if (any_cs && !(ws_ok && need <= g_ws_bytes && fast_ok(L, true) && (g_gemm_mode != 1 || L.p[0].elt))) any_cs = false;
      if (want_cs && !any_cs) {
        // this launch cannot carry its column sums (no workspace, generic kernel): fp32 operands get them from the streaming
        // kernel right here, so that a Batch whose problems span several launches (h = 768: 21 column-block problems) never ends
        // up with some bias gradients fused and others missing; bf16 operands have no such kernel -- the caller is told
        for (int i = 0; i < L.nprob; ++i) {
          if (!cs_out[i]) continue;
          const Seg& sg = L.p[i].seg[0];
          if (L.p[i].elt || sg.lda != L.p[i].M) { colsum_missed = true; continue; }
          if (launch_colsum3(sg.A, nullptr, nullptr, cs_out[i], nullptr, nullptr, sg.K, L.p[i].M, s, cs_out2[i], nullptr, nullptr)) colsum_missed = true;
          else colsum_fused = true;
        }
      }
      if (ws_ok && need <= g_ws_bytes) {
        ReduceArgs R;
        R.n = L.nprob;
        float* w = g_ws;
        int max_elems = 0;
        for (int i = 0; i < L.nprob; ++i) {
          Problem& q = L.p[i];
          R.it[i] = ReduceItem{w, q.C, q.M, q.N, q.ldc, L.ksplit, (long long)q.M * q.N};
          q.C = w; q.ldc = q.N; q.epi = EPI_STORE; q.accumulate = 0; q.split_stride = (long long)q.M * q.N;
          w += (size_t)L.ksplit * q.M * q.N;
          if (q.M * (q.N / 4) > max_elems) max_elems = q.M * (q.N / 4);
        }
        ReduceArgs RC;          // bias-gradient partials [ksplit][M] behind the tiles, summed into one or two outputs
        RC.n = 0;
        int max_cs = 0;
        for (int i = 0; i < L.nprob && any_cs; ++i) {
          Problem& q = L.p[i];
          if (!cs_out[i]) continue;
          q.colsum = w; q.colsum_stride = q.M;
          RC.it[RC.n++] = ReduceItem{w, cs_out[i], 1, q.M, q.M, L.ksplit, (long long)q.M};
          if (cs_out2[i] && RC.n < GH_MAX_PROBLEMS) RC.it[RC.n++] = ReduceItem{w, cs_out2[i], 1, q.M, q.M, L.ksplit, (long long)q.M};
          w += (size_t)L.ksplit * q.M;
          if (q.M / 4 > max_cs) max_cs = q.M / 4;
        }
        hipError_t e = launch_any();
        if (e != hipSuccess) err = e;
        // one reduce launch: the bias-gradient partials ride behind the tiles (their workgroups beyond the first exit at once)
        for (int i = 0; i < RC.n; ++i) R.it[R.n++] = RC.it[i];
        if (max_cs > max_elems) max_elems = max_cs;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((max_elems + 255) / 256, R.n), dim3(256), 0, s, R);
        if (RC.n > 0) colsum_fused = true;
        e = hipGetLastError();
        if (e != hipSuccess) err = e;
        reset();
        return;
      }
    }
    if (tn && tn_pp) { err = hipErrorInvalidValue; reset(); return; }      // (the ping-pong kernel writes partial tiles only: no workspace, no launch)
    if (!tn && nt_split_plan()) return;
    hipError_t e = launch_any();
    if (e != hipSuccess) err = e;
    reset();
  }

  hipError_t launch_any() {
    if (tn && tn_pp) return launch_tn_pp(L, s);
#ifdef GH_MEASURE      // (the fp32 ping-pong tile is a measured-and-dropped experiment, DESIGN 4.5: instantiated in the tool build only)
    if (pp32 && !tn && L.ksplit == 1 && !L.p[0].elt) return launch_cfg<4, 2, 5>(L, tn, s);
#endif
    if (narrow && !tn && L.ksplit == 1 && !L.p[0].elt) return launch_cfg<2, 2, 5>(L, tn, s);
    if (big && !tn && L.ksplit == 1 && !L.p[0].elt) {
      // Occupancy-aware tile choice.  The chip holds 768 workgroups of either configuration (3 per CU); a grid of 64-row
      // tiles that ends in a thinly filled last round (e.g. 976 workgroups = 1.27 rounds, the node-compact single-problem
      // launches) runs faster on 32-row tiles (1937 workgroups = 2.52 rounds): measured 83 vs 78 TF at K = 300.
      static int mode = -1;
      if (mode < 0) mode = measure_env("GH_NT_TILE_RULE", 1);
      const double r = (double)L.m_tiles * L.nprob / 768.0;
      const double frac = r - (double)(long long)r;
      if (mode == 1 && g_gemm_mode != 1 && r < 3.0 && frac > 0.02 && frac < 0.45) {
        int mt = 0;
        for (int i = 0; i < L.nprob; ++i) { const int t = (L.p[i].M + 31) / 32; if (t > mt) mt = t; }
        L.m_tiles = mt;
        return launch_cfg<1, 4, 5>(L, tn, s);
      }
    }
    if (wide256) {
      // the ping-pong loop's preconditions (gemm_nt_pp.hip.h): whole 64-deep K tiles, no K split, no dropout inside the K loop
      bool ok = L.ksplit == 1;
      for (int i = 0; i < L.nprob; ++i) {
        if (L.p[i].drop_mode == 1) ok = false;
        for (int j = 0; j < L.p[i].nseg; ++j) if (L.p[i].seg[j].K % 64) ok = false;
      }
      if (ok) return launch_cfg<2, 4, 4, 8>(L, tn, s);
      int mt = 0;
      for (int i = 0; i < L.nprob; ++i) { const int t = (L.p[i].M + 127) / 128; if (t > mt) mt = t; }
      L.m_tiles = mt;
      return launch_cfg<2, 2, 8, 4>(L, tn, s);
    }
    if (wide128) return launch_cfg<2, 2, 4, 4>(L, tn, s);
    if (wide) return launch_cfg<2, 2, 8, 4>(L, tn, s);
    return big ? launch_cfg<2, 2, 10>(L, tn, s) : launch_cfg<1, 4, 5>(L, tn, s);
  }

  // Few-row NT GEMMs (evidence level, head): too few row tiles to fill 256 CUs, so split K across
  // workgroups, keep the partial tiles in the workspace and finish with nt_finish_kernel.
  bool nt_split_plan() {
    const int blocks = L.m_tiles * L.nprob;
    static int max_blocks = -1;
    if (max_blocks < 0) max_blocks = measure_env("GH_NT_SPLIT_BLOCKS", 96);
    if (g_ws == nullptr || blocks >= max_blocks || !fast_ok(L, false)) return false;
    int tmax = 0;          // K tiles over the concatenated segments (every problem of a launch is split alike)
    for (int i = 0; i < L.nprob; ++i) {
      const Problem& q = L.p[i];
      if (q.epi == EPI_ATOMIC || q.epi == EPI_GATE_PRE || q.drop_mode != 0 || (q.epi == EPI_TANH_H && q.w2) || q.seg0_rows > 0 || q.elt) return false;   // (fp32 problems only)
      int t = 0;
      for (int j = 0; j < q.nseg; ++j) t += (q.seg[j].K + 15) / 16;
      if (i > 0 && t != tmax) return false;
      tmax = t;
    }
    int ks = tmax / 4;
    // workgroups wanted: one per CU -- two for long contractions (>= 128 K tiles, e.g. the evidence-level attention at h = 768:
    // K = 6272), where a split-K workgroup is MFMA-bound rather than latency-bound (A/B configs[4] bf16: 256 / 512 / 768 =
    // 107.7 / 108.7 / 108.1 K pairs/s; headline and Snopes-count steps unchanged at any of them)
    static int target = -1;
    if (target < 0) target = measure_env("GH_NT_SPLIT_TARGET", 0);
    const int tgt = target > 0 ? target : (tmax >= 128 ? 512 : 256);
    const int want = (tgt + blocks - 1) / blocks;
    if (ks > want) ks = want;
    if (ks < 2) return false;
    const int ct = (tmax + ks - 1) / ks;     // tiles per chunk
    ks = (tmax + ct - 1) / ct;
    if (ks < 2) return false;
    size_t need = 0;
    for (int i = 0; i < L.nprob; ++i) need += (size_t)ks * L.p[i].M * L.p[i].N * sizeof(float);
    if (need > g_ws_bytes) return false;
    FinishArgs F;
    F.n = L.nprob;
    float* w = g_ws;
    int max_m = 0;
    for (int i = 0; i < L.nprob; ++i) {
      Problem& q = L.p[i];
      F.it[i] = FinishItem{w, (long long)q.M * q.N, ks, q.M, q.N, q.epi, q.accumulate, q.ldc,
                           q.C, q.bias, q.bias2, q.out1, q.in0, q.in1, q.u, q.w2, q.e, q.ldu, q.R, q.heads, q.rowg};
      q.C = w; q.ldc = q.N; q.epi = EPI_STORE; q.accumulate = 0; q.bias = nullptr; q.bias2 = nullptr;
      q.split_stride = (long long)q.M * q.N;
      w += (size_t)ks * q.M * q.N;
      if (q.M > max_m) max_m = q.M;
    }
    L.ksplit = ks;
    L.kchunk = ct * 16;
    hipError_t e = launch_any();
    if (e != hipSuccess) err = e;
    // many partials over few rows: a workgroup per row (the four waves share the partials); otherwise a wave per row
    F.split = (ks >= 16 && max_m <= 8192) ? 1 : 0;      // (ks = 9, M = 960 -- the claim cell, the evidence-level attention -- measured equal or slower split)
    hipLaunchKernelGGL(nt_finish_kernel, dim3(F.split ? max_m : (max_m + 3) / 4, F.n), dim3(256), 0, s, F);
    split_done = true;
    e = hipGetLastError();
    if (e != hipSuccess) err = e;
    reset();
    return true;
  }
};

static Problem gemm_problem(int M, int N, int epi, float* C, int ldc, const float* A, int lda, const float* B, int ldb,
                            int K, const int32_t* gatherA = nullptr, int elt = 0) {
  Problem p = make_problem(M, N, epi, C, ldc);
  p.elt = elt;
  p.seg[0] = make_seg_nt(A, lda, B, ldb, K, gatherA, elt);
  return p;
}
static void add_seg(Problem& p, const float* A, int lda, const float* B, int ldb, int K) {
  p.seg[p.nseg++] = make_seg_nt(A, lda, B, ldb, K, nullptr, p.elt);
}
static void set_dropout(Problem& p, int mode, int ld, float drop_p, unsigned seed) {
  if (drop_p <= 0.f) return;
  p.drop_mode = mode; p.drop_ld = ld; p.drop_col0 = 0; p.drop_seed = seed;
  const double t = (double)drop_p * 4294967296.0;
  p.drop_thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
  p.drop_scale = 1.0f / (1.0f - drop_p);
}
static Problem tn_problem(int I, int J, float* C, int ldc, const float* A, int lda, const float* B, int ldb, int K,
                          const int32_t* gatherB = nullptr, int elt = 0) {
  Problem p = make_problem(I, J, EPI_ATOMIC, C, ldc);
  p.seg[0] = make_seg_tn(A, lda, I, B, ldb, J, K, nullptr, gatherB);
  p.elt = elt;
  if (elt) {
    p.seg[0].vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && lda % 8 == 0 && I % 8 == 0;
    p.seg[0].vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && ldb % 8 == 0 && J % 8 == 0;
  }
  return p;
}

}  // namespace gh

using namespace gh;

// bf: bf16 storage pipeline -- x (or the table behind ids), the weights and xp/a/z/rr/rx/hh/out hold bf16 (pointers typed
// float* all the same); out32 then receives the fp32 copy of the cell output.  Biases, score_w and score_x stay fp32.
int gh::cell_fwd_impl(int bf, float* out32, const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                                const int32_t* goff, int m_real, int m_rows,
                                const float* x, const int32_t* ids, int n, int r, int din, int h,
                                const float* w_p, const float* w_z0, const float* w_z1, const float* w_r0,
                                const float* w_r1, const float* w_h0, const float* w_h1,
                                const float* b_z0, const float* b_z1, const float* b_r0, const float* b_r1,
                                const float* b_h0, const float* b_h1,
                                float* xp, float* a, float* z, float* rr, float* rx, float* hh, float* out,
                                float drop_p, uint32_t drop_seed,
                                const float* score_w, float* score_x, float score_drop_p, uint32_t score_drop_seed,
                                gh_stream_t stream, int pad_out_dead, int* score_parts, float* xdrop) {
  hipStream_t s = (hipStream_t)stream;
  if (score_parts) *score_parts = 1;
  GH_REQUIRE(n > 0 && r > 0 && din > 0 && h > 0, "ggnn_cell_fwd: bad sizes n=%d r=%d din=%d h=%d", n, r, din, h);
  // (out32 may be NULL when no consumer reads the cell output as fp32: composite forward -- the first cell's scorer projection is fused,
  //  the second cell's word attention reads the bf16 rows)
  GH_REQUIRE(!bf || (din % 8 == 0 && h % 8 == 0), "ggnn_cell_fwd_bf16: needs din %% 8 == 0, h %% 8 == 0 (din=%d h=%d)", din, h);
  if (!goff) { m_real = n * r; m_rows = n * r; }
  GH_REQUIRE(m_real >= 0 && m_real <= m_rows && m_rows <= n * r, "ggnn_cell_fwd: node-compact rows %d/%d do not fit n*r=%d", m_real, m_rows, n * r);
  const int M = m_rows;
  if (M == 0) return 0;
  GH_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "ggnn_cell_fwd: dropout p=%f not in [0,1)", drop_p);
  // the stateless dropout mask hashes a 32-BIT element index row * width + column (gemm.hip.h drop_hash): beyond 2^32 elements
  // two rows would share their mask
  GH_REQUIRE((drop_p <= 0.f && score_drop_p <= 0.f) || (long long)M * (long long)(din > h ? din : h) < (1LL << 32),
             "ggnn_cell_fwd: %d rows x %d columns exceed the dropout mask's 32-bit element index", M, din > h ? din : h);
  GH_REQUIRE((score_w == nullptr) == (score_x == nullptr), "ggnn_cell_fwd: score_w and score_x come together");
  GH_REQUIRE(score_drop_p >= 0.f && score_drop_p < 1.f, "ggnn_cell_fwd: scorer dropout p=%f not in [0,1)", score_drop_p);
  const bool wide = bf && h % 256 == 0;      // 128 x 256 bf16 tiles cover the width exactly (h = 768)
  // bf16 storage with dropout: masking the A fragments costs ~100 VALU instructions per 16-row tile and K tile (eight hashes on
  // packed bf16 pairs) against 8 MFMAs -- the projection ran at half the rate of the same product without a mask (configs[4]:
  // 376 us against ~190).  With a scratch row buffer from the caller (xdrop [m_rows][din] bf16) the masked operand -- gathered
  // embedding rows included -- is materialised once by the streaming kernel the backward used anyway (launch_gather_rows), the
  // projection reads it as a plain operand, and the backward's dW_proj re-uses it instead of gathering again.
  const bool pre_drop = bf && xdrop && drop_p > 0.f && din <= h;
  if (pre_drop) {
    if (int e = launch_gather_rows(x, ids, xdrop, M, din, s, drop_p, drop_seed, 1)) return e;
  }
  {  // xp = dropout(x) Wp^T   (wrapper.py:189-191); embedding rows gathered by the loader, the dropout mask applied to the fragments
    Batch b(false, M, s, wide, 7, h);
    Problem p = pre_drop ? gemm_problem(M, h, EPI_STORE, xp, h, xdrop, din, w_p, din, din, nullptr, bf)
                         : gemm_problem(M, h, EPI_STORE, xp, h, x, din, w_p, din, din, ids, bf);
    p.io = bf ? 1 : 0;
    if (!pre_drop) set_dropout(p, 1, din, drop_p, drop_seed);
    b.add(p);
    b.flush();
    GH_REQUIRE(b.err != hipErrorInvalidValue || drop_p == 0.f, "ggnn_cell_fwd: fused dropout needs float4-shaped rows (din=%d, h=%d)", din, h);
    GH_CHECK_HIP(b.err);
  }
  bool partial_zero = false;
  // zero fills of this cell -- the padding rows of `a` that a row tile can touch (below) and the scorer's partial dot products
  // (two column blocks add into score_x) -- ride on the aggregation launch when its kernel variant takes them: two memset
  // dispatches less per step
  ZeroFill zf = {nullptr, 0, nullptr, 0};
  bool zf_done = false, score_zero = false;
  if (score_w) {
    Batch bh(false, M, s, false, 2, h);
    score_zero = bh.narrow && h <= 2 * bh.bn && h > bh.bn;
  }
  if (m_rows > m_real) {
    // padding rows of the node-compact layout have no neighbours.  The fast gate GEMMs skip the aggregation segment for
    // every row tile at or beyond seg0_rows (= m_real), so only the tile that straddles m_real ever reads such rows: zero
    // the rows up to the next 128-row boundary (the largest row tile) plus one tile instead of all (m_rows - m_real) of
    // them (40 MB at the bench shape).  Only when the operands are shaped for the fast kernel -- the generic kernel
    // reads every row; the launches below are checked to have taken the fast path.
    const uintptr_t al = (uintptr_t)a | (uintptr_t)xp | (uintptr_t)z | (uintptr_t)rr | (uintptr_t)rx | (uintptr_t)hh | (uintptr_t)out |
                         (uintptr_t)w_z0 | (uintptr_t)w_z1 | (uintptr_t)w_r0 | (uintptr_t)w_r1 | (uintptr_t)w_h0 | (uintptr_t)w_h1 |
                         (uintptr_t)b_z0 | (uintptr_t)b_z1 | (uintptr_t)b_r0 | (uintptr_t)b_r1 | (uintptr_t)b_h0 | (uintptr_t)b_h1;
    // (and only while the activations stay below the 2 GB a buffer descriptor addresses: beyond that the gate GEMMs take the
    //  generic kernel -- fast_ok -- which reads every row)
    partial_zero = ((al & 15) == 0) && (h % (bf ? 8 : 4) == 0) && h >= 4 && 4.0 * (double)h * (double)M < 2147483648.0;
    const int zfull = ((m_real + 127) / 128 + 1) * 128;
    const int zend = (partial_zero && zfull < m_rows) ? zfull : m_rows;
    if (!bf) { zf.p0 = a + (size_t)m_real * h; zf.n0 = (long long)(zend - m_real) * h; }
    if (!bf && score_zero && M % 4 == 0) { zf.p1 = score_x; zf.n1 = M; }
    if (int e = launch_spmm(bits, dinv, vals, keep, goff, m_real, xp, a, n, r, h, 0, 0, s, bf, bf ? nullptr : &zf, &zf_done)) return e;   // a = A_hat xp (:192)
    if (!zf_done)
      GH_CHECK_HIP(hipMemsetAsync((char*)a + (size_t)m_real * h * (bf ? 2 : 4), 0, (size_t)(bf ? 2 : 4) * (size_t)(zend - m_real) * h, s));
  } else {
    if (int e = launch_spmm(bits, dinv, vals, keep, goff, m_real, xp, a, n, r, h, 0, 0, s, bf)) return e;   // a = A_hat xp (:192)
  }
  const long long generic_before = g_path_counts[1];
  {  // z, r gates (:194-200): [a | xp] . [W?0 | W?1]^T as two K segments
    Batch b(false, M, s, wide, 1, h);
    Problem pz = gemm_problem(M, h, EPI_SIGMOID_Z, z, h, a, h, w_z0, h, h, nullptr, bf);
    pz.io = bf ? 1 : 0;
    add_seg(pz, xp, h, w_z1, h, h);
    pz.bias = b_z0; pz.bias2 = b_z1;
    Problem pr = gemm_problem(M, h, EPI_SIGMOID_R, rr, h, a, h, w_r0, h, h, nullptr, bf);
    pr.io = bf ? 7 : 0;
    add_seg(pr, xp, h, w_r1, h, h);
    pr.bias = b_r0; pr.bias2 = b_r1; pr.out1 = rx; pr.in0 = xp;
    if (m_rows > m_real) pz.seg0_rows = pr.seg0_rows = (m_real > 0 ? m_real : 1);
    b.add(pz); b.add(pr);
    b.flush();
    GH_CHECK_HIP(b.err);
  }
  {  // h gate and the convex update (:202-206); optionally the GSL word scorer's projection of the result (:167)
    // score_parts (caller provides score_x for ceil(h / column block) x M floats): rows wider than two column blocks keep their
    // tile and every block writes its share of the projection to a partial of its own ([block][M]; gh_scorer_gsl score_parts)
    const bool parts_ok = score_w && score_parts != nullptr;
    Batch b(false, M, s, wide && (!score_w || parts_ok), 2, h);
    if (parts_ok && h > b.bn && !(b.narrow && h <= 2 * b.bn)) {
      b.e_block_stride = M;
      *score_parts = (h + b.bn - 1) / b.bn;
    } else if (score_w && b.narrow) {
      if (h > 2 * b.bn) { b.narrow = false; b.bn = 320; }      // more than two column blocks: whole rows on the 320-wide tile
      else if (h > b.bn && !(zf_done && zf.p1 == score_x)) GH_CHECK_HIP(hipMemsetAsync(score_x, 0, sizeof(float) * (size_t)M, s));   // two blocks add their partial dot products
    }
    Problem ph = gemm_problem(M, h, EPI_TANH_H, hh, h, a, h, w_h0, h, h, nullptr, bf);
    ph.io = bf ? 15 : 0; ph.c32 = bf ? out32 : nullptr;
    add_seg(ph, rx, h, w_h1, h, h);
    ph.bias = b_h0; ph.bias2 = b_h1; ph.out1 = out; ph.in0 = z; ph.in1 = xp;
    if (m_rows > m_real) ph.seg0_rows = (m_real > 0 ? m_real : 1);
    if (score_w) {
      ph.w2 = score_w; ph.e = score_x;
      if (pad_out_dead && m_rows > m_real) ph.ldu = 1;
      set_dropout(ph, 2, h, score_drop_p, score_drop_seed);
    }
    b.add(ph);
    b.flush();
    GH_REQUIRE(b.err != hipErrorInvalidValue || !score_w, "ggnn_cell_fwd: the fused scorer projection needs h %% 4 == 0 and h <= 320 (h=%d)", h);
    GH_CHECK_HIP(b.err);
  }
  GH_REQUIRE(!partial_zero || g_path_counts[1] == generic_before,
             "ggnn_cell_fwd: internal -- a gate GEMM took the generic kernel although the padding rows of `a` were only zeroed for the fast one");
  return 0;
}

extern "C" int gh_ggnn_cell_fwd(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                                const int32_t* goff, int m_real, int m_rows,
                                const float* x, const int32_t* ids, int n, int r, int din, int h,
                                const float* w_p, const float* w_z0, const float* w_z1, const float* w_r0,
                                const float* w_r1, const float* w_h0, const float* w_h1,
                                const float* b_z0, const float* b_z1, const float* b_r0, const float* b_r1,
                                const float* b_h0, const float* b_h1,
                                float* xp, float* a, float* z, float* rr, float* rx, float* hh, float* out,
                                float drop_p, uint32_t drop_seed,
                                const float* score_w, float* score_x, float score_drop_p, uint32_t score_drop_seed,
                                gh_stream_t stream) {
  return cell_fwd_impl(0, nullptr, bits, dinv, vals, keep, goff, m_real, m_rows, x, ids, n, r, din, h, w_p, w_z0, w_z1, w_r0, w_r1,
                       w_h0, w_h1, b_z0, b_z1, b_r0, b_r1, b_h0, b_h1, xp, a, z, rr, rx, hh, out, drop_p, drop_seed, score_w,
                       score_x, score_drop_p, score_drop_seed, stream, 0, nullptr, nullptr);
}

extern "C" int gh_ggnn_cell_fwd_bf16(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                                     const int32_t* goff, int m_real, int m_rows,
                                     const void* x, const int32_t* ids, int n, int r, int din, int h,
                                     const void* w_p, const void* w_z0, const void* w_z1, const void* w_r0,
                                     const void* w_r1, const void* w_h0, const void* w_h1,
                                     const float* b_z0, const float* b_z1, const float* b_r0, const float* b_r1,
                                     const float* b_h0, const float* b_h1,
                                     void* xp, void* a, void* z, void* rr, void* rx, void* hh, void* out, float* out32,
                                     float drop_p, uint32_t drop_seed,
                                     const float* score_w, float* score_x, float score_drop_p, uint32_t score_drop_seed,
                                     gh_stream_t stream) {
  typedef const float* cf;
  typedef float* mf;
  return cell_fwd_impl(1, out32, bits, dinv, vals, keep, goff, m_real, m_rows, (cf)x, ids, n, r, din, h, (cf)w_p, (cf)w_z0, (cf)w_z1,
                       (cf)w_r0, (cf)w_r1, (cf)w_h0, (cf)w_h1, b_z0, b_z1, b_r0, b_r1, b_h0, b_h1, (mf)xp, (mf)a, (mf)z, (mf)rr,
                       (mf)rx, (mf)hh, (mf)out, drop_p, drop_seed, score_w, score_x, score_drop_p, score_drop_seed, stream, 0, nullptr, nullptr);
}

// bf: bf16 storage pipeline -- x / table, wt_*, the saved xp..hh and the scratch dhp..da hold bf16; g, dx and every weight /
// bias gradient stay fp32.
int gh::cell_bwd_impl(int bf, const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                                const int32_t* goff, int m_real,
                                const float* x, const int32_t* ids, int n, int r, int din, int h,
                                const float* wt_p, const float* wt_z0, const float* wt_z1, const float* wt_r0,
                                const float* wt_r1, const float* wt_h0, const float* wt_h1,
                                const float* xp, const float* a, const float* z, const float* rr, const float* rx,
                                const float* hh, const float* g,
                                float* dhp, float* dzp, float* drp, float* dxp, float* da,
                                float* dx, float* dw_p, float* dw_z0, float* dw_z1, float* dw_r0, float* dw_r1,
                                float* dw_h0, float* dw_h1, float* db_z, float* db_r, float* db_h,
                                float* db_z1, float* db_r1, float* db_h1, float drop_p, uint32_t drop_seed,
                                gh_stream_t stream, gh_stream_t wstream, hipEvent_t ev_l1, hipEvent_t ev_agg,
                                int pre_done, const GateFuse* next, const float* xdrop) {
  hipStream_t s = (hipStream_t)stream;
  // Weight-gradient stream (composite backward, model_ops.hip): the split-K weight-gradient GEMMs only need dzp / drp / dhp
  // (final after the first dX launch) and, for dW_proj, dxp (final after the aggregation).  On their own stream they run
  // underneath the rest of the chain -- the streaming kernels of one stream (gate_bwd_pre, aggregation, gather, partial
  // reduces) hide under the other stream's MFMA-bound launches, which two launches on ONE stream never do.
  hipStream_t sw = wstream ? (hipStream_t)wstream : s;
  const bool two = sw != s;
  GH_REQUIRE(!two || (ev_l1 && ev_agg), "ggnn_cell_bwd: a separate weight-gradient stream needs its two events");
  GH_REQUIRE(n > 0 && r > 0 && din > 0 && h > 0, "ggnn_cell_bwd: bad sizes");
  GH_REQUIRE(!bf || (din % 8 == 0 && h % 8 == 0 && din <= h), "ggnn_cell_bwd_bf16: needs din %% 8 == 0, h %% 8 == 0, din <= h (din=%d h=%d)", din, h);
  if (!goff) m_real = n * r;
  GH_REQUIRE(m_real >= 0 && m_real <= n * r, "ggnn_cell_bwd: node-compact rows %d do not fit n*r=%d", m_real, n * r);
  GH_REQUIRE(drop_p <= 0.f || (long long)n * r * (long long)din < (1LL << 32),
             "ggnn_cell_bwd: %d rows x %d columns exceed the dropout mask's 32-bit element index", n * r, din);
  const int M = m_real;      // padding rows receive no gradient and contribute none
  if (M == 0) return 0;
  // out = h z + xp (1-z):  dhp = g z (1-h^2), dzp = g (h-xp) z (1-z), dxp = g (1-z)
  // (pre_done: the GEMM that produced g already wrote the three straight from its epilogue, EPI_GATE_PRE -- g itself was
  //  never stored and gate_bwd_pre's 4 reads + 3 writes shrink to the 3 + 3 the producing epilogue added)
  if (!pre_done)
    if (int e = launch_gate_bwd_pre(g, z, hh, xp, dhp, dzp, dxp, (size_t)M * h, s, bf)) return e;
  const bool wide = bf && h % 256 == 0;
  {  // hp = a Wh0^T + (r xp) Wh1^T:  da = dhp Wh0 ; d(r xp) = dhp Wh1 -> drp, dxp += .
    Batch b(false, M, s, wide, 3, h);
    Problem p0 = gemm_problem(M, h, EPI_STORE, da, h, dhp, h, wt_h0, h, h, nullptr, bf);
    Problem p1 = gemm_problem(M, h, EPI_BWD_DRX, drp, h, dhp, h, wt_h1, h, h, nullptr, bf);
    p1.out1 = dxp; p1.in0 = xp; p1.in1 = rr;
    p0.io = bf ? 1 : 0; p1.io = bf ? 15 : 0;
    b.add(p0); b.add(p1);
    b.flush();
    GH_CHECK_HIP(b.err);
  }
  const bool cs = bf ? true : (h % 4 == 0);      // (any float4-shaped width since round 5: Batch::flush satisfies every request itself)
  bool colsum_done = false;
  if (two) {  // six of the seven weight gradients start here, on the weight-gradient stream
    GH_CHECK_HIP(hipEventRecord(ev_l1, s));
    GH_CHECK_HIP(hipStreamWaitEvent(sw, ev_l1, 0));
    Batch b(true, M, sw);
    b.add(tn_problem(h, h, dw_z0, h, dzp, h, a, h, M, nullptr, bf));  if (cs) b.want_colsum(db_z, db_z1);
    b.add(tn_problem(h, h, dw_z1, h, dzp, h, xp, h, M, nullptr, bf));
    b.add(tn_problem(h, h, dw_r0, h, drp, h, a, h, M, nullptr, bf));  if (cs) b.want_colsum(db_r, db_r1);
    b.add(tn_problem(h, h, dw_r1, h, drp, h, xp, h, M, nullptr, bf));
    b.add(tn_problem(h, h, dw_h0, h, dhp, h, a, h, M, nullptr, bf));  if (cs) b.want_colsum(db_h, db_h1);
    b.add(tn_problem(h, h, dw_h1, h, dhp, h, rx, h, M, nullptr, bf));
    b.flush();
    GH_CHECK_HIP(b.err);
    colsum_done = b.colsum_fused && !b.colsum_missed;
    if (!colsum_done) {
      GH_REQUIRE(!bf, "ggnn_cell_bwd_bf16: the bias gradients need the split-K workspace (gh_set_workspace)");
      if (int e = launch_colsum3(dzp, drp, dhp, db_z, db_r, db_h, M, h, sw, db_z1, db_r1, db_h1)) return e;
    }
  }
  {  // da += dzp Wz0 + drp Wr0 ; dxp += dzp Wz1 + drp Wr1
    Batch b(false, M, s, wide, 4, h);
    Problem p0 = gemm_problem(M, h, EPI_STORE, da, h, dzp, h, wt_z0, h, h, nullptr, bf);
    add_seg(p0, drp, h, wt_r0, h, h);
    p0.accumulate = 1;
    Problem p1 = gemm_problem(M, h, EPI_STORE, dxp, h, dzp, h, wt_z1, h, h, nullptr, bf);
    add_seg(p1, drp, h, wt_r1, h, h);
    p1.accumulate = 1;
    p0.io = p1.io = bf ? 1 : 0;
    b.add(p0); b.add(p1);
    b.flush();
    GH_CHECK_HIP(b.err);
  }
  if (int e = launch_spmm(bits, dinv, vals, keep, goff, m_real, da, dxp, n, r, h, 1, 1, s, bf)) return e;   // dxp += A_hat^T da
  if (two) {
    GH_CHECK_HIP(hipEventRecord(ev_agg, s));
    GH_CHECK_HIP(hipStreamWaitEvent(sw, ev_agg, 0));
  }
  if (dx || next) {  // dx = (dxp Wp) . mask/(1-p)
    GH_REQUIRE(!next || (din % 4 == 0 && (next->bf16 != 0) == (bf != 0)), "ggnn_cell_bwd: the fused gate head needs float4-shaped rows and both cells in the same storage mode");
    Batch b(false, M, s, wide && din % 256 == 0, 5, din);
    Problem p = gemm_problem(M, din, next ? EPI_GATE_PRE : EPI_STORE, next ? next->dhp : dx, din, dxp, h, wt_p, h, h, nullptr, bf);      // dx itself is fp32
    if (next) {      // dx IS the gradient w.r.t. the previous cell's output: write that cell's dhp / dzp / dxp instead of dx
      p.in0 = next->z; p.in1 = next->hh; p.in2 = next->xp; p.out1 = next->dzp; p.out2 = next->dxp;
      p.io = next->bf16 ? (1 | 2 | 4 | 8 | 16 | 32) : 0;
    }
    set_dropout(p, 3, din, drop_p, drop_seed);
    b.add(p);
    b.flush();
    GH_CHECK_HIP(b.err);
  }
  {  // weight gradients: G^T X over the M rows, split-K partial tiles + reduce.  The bias gradients are the column
     // sums of dzp / drp / dhp: they ride along with the first GEMM that streams each of them (no separate
     // column-sum pass over 3 x M x h values) whenever the launch takes the workspace path.
    Batch b(true, M, sw);
    if (!two) {
      b.add(tn_problem(h, h, dw_z0, h, dzp, h, a, h, M, nullptr, bf));  if (cs) b.want_colsum(db_z, db_z1);
      b.add(tn_problem(h, h, dw_z1, h, dzp, h, xp, h, M, nullptr, bf));
      b.add(tn_problem(h, h, dw_r0, h, drp, h, a, h, M, nullptr, bf));  if (cs) b.want_colsum(db_r, db_r1);
      b.add(tn_problem(h, h, dw_r1, h, drp, h, xp, h, M, nullptr, bf));
      b.add(tn_problem(h, h, dw_h0, h, dhp, h, a, h, M, nullptr, bf));  if (cs) b.want_colsum(db_h, db_h1);
      b.add(tn_problem(h, h, dw_h1, h, dhp, h, rx, h, M, nullptr, bf));
    }
    if (bf && xdrop && drop_p > 0.f && din <= h) {
      // the forward left the masked (and gathered) operand rows in xdrop (cell_fwd_impl pre_drop): no second gather pass
      b.add(tn_problem(h, din, dw_p, din, dxp, h, xdrop, din, M, nullptr, bf));
    } else if ((ids || drop_p > 0.f || bf) && din <= h) {
      // operand rows materialised once into the (now free) `da` scratch -- embedding gather and/or the forward's
      // dropout mask applied in that one streaming pass -- so the split-K GEMM loader stays a plain copy
      if (int e = launch_gather_rows(x, ids, da, M, din, sw, drop_p, drop_seed, bf)) return e;
      b.add(tn_problem(h, din, dw_p, din, dxp, h, da, din, M, nullptr, bf));
    } else {
      GH_REQUIRE(drop_p == 0.f, "ggnn_cell_bwd: fused dropout needs din (%d) <= h (%d)", din, h);
      b.add(tn_problem(h, din, dw_p, din, dxp, h, x, din, M, ids));
    }
    b.flush();
    GH_CHECK_HIP(b.err);
    if (two || (b.colsum_fused && !b.colsum_missed)) return 0;
  }
  GH_REQUIRE(!bf, "ggnn_cell_bwd_bf16: the bias gradients need the split-K workspace (gh_set_workspace)");
  return launch_colsum3(dzp, drp, dhp, db_z, db_r, db_h, M, h, s, db_z1, db_r1, db_h1);
}

extern "C" int gh_ggnn_cell_bwd(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                                const int32_t* goff, int m_real,
                                const float* x, const int32_t* ids, int n, int r, int din, int h,
                                const float* wt_p, const float* wt_z0, const float* wt_z1, const float* wt_r0,
                                const float* wt_r1, const float* wt_h0, const float* wt_h1,
                                const float* xp, const float* a, const float* z, const float* rr, const float* rx,
                                const float* hh, const float* g,
                                float* dhp, float* dzp, float* drp, float* dxp, float* da,
                                float* dx, float* dw_p, float* dw_z0, float* dw_z1, float* dw_r0, float* dw_r1,
                                float* dw_h0, float* dw_h1, float* db_z, float* db_r, float* db_h,
                                float* db_z1, float* db_r1, float* db_h1, float drop_p, uint32_t drop_seed,
                                gh_stream_t stream) {
  return cell_bwd_impl(0, bits, dinv, vals, keep, goff, m_real, x, ids, n, r, din, h, wt_p, wt_z0, wt_z1, wt_r0, wt_r1, wt_h0, wt_h1,
                       xp, a, z, rr, rx, hh, g, dhp, dzp, drp, dxp, da, dx, dw_p, dw_z0, dw_z1, dw_r0, dw_r1, dw_h0, dw_h1,
                       db_z, db_r, db_h, db_z1, db_r1, db_h1, drop_p, drop_seed, stream, nullptr, nullptr, nullptr, 0, nullptr, nullptr);
}

extern "C" int gh_ggnn_cell_bwd_bf16(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                                     const int32_t* goff, int m_real,
                                     const void* x, const int32_t* ids, int n, int r, int din, int h,
                                     const void* wt_p, const void* wt_z0, const void* wt_z1, const void* wt_r0,
                                     const void* wt_r1, const void* wt_h0, const void* wt_h1,
                                     const void* xp, const void* a, const void* z, const void* rr, const void* rx,
                                     const void* hh, const float* g,
                                     void* dhp, void* dzp, void* drp, void* dxp, void* da,
                                     float* dx, float* dw_p, float* dw_z0, float* dw_z1, float* dw_r0, float* dw_r1,
                                     float* dw_h0, float* dw_h1, float* db_z, float* db_r, float* db_h,
                                     float* db_z1, float* db_r1, float* db_h1, float drop_p, uint32_t drop_seed,
                                     gh_stream_t stream) {
  typedef const float* cf;
  typedef float* mf;
  return cell_bwd_impl(1, bits, dinv, vals, keep, goff, m_real, (cf)x, ids, n, r, din, h, (cf)wt_p, (cf)wt_z0, (cf)wt_z1, (cf)wt_r0,
                       (cf)wt_r1, (cf)wt_h0, (cf)wt_h1, (cf)xp, (cf)a, (cf)z, (cf)rr, (cf)rx, (cf)hh, g, (mf)dhp, (mf)dzp, (mf)drp,
                       (mf)dxp, (mf)da, dx, dw_p, dw_z0, dw_z1, dw_r0, dw_r1, dw_h0, dw_h1, db_z, db_r, db_h, db_z1, db_r1,
                       db_h1, drop_p, drop_seed, stream, nullptr, nullptr, nullptr, 0, nullptr, nullptr);
}

// Concat attention, generalised for the composite model entry points (model_ops.hip):
//   nl / rowu: `left` has nl rows and output row m takes the left projection u[rowu[m]] (the word level's left input is the
//              CLAIM vector: one u row per claim instead of one per pair; rowu = NULL keeps nl == b and the per-pair map);
//   att_out / att_ld: where `attended` goes (row pitch att_ld >= dr * heads: straight into a wider concatenation buffer).
int gh::att_fwd_impl(const float* left, int nl, const int32_t* rowu, const float* right, const float* mask, const int32_t* goff,
                 const int32_t* rowg, int m_real, int b, int l, int xl, int dr, int ha, int heads, const float* w1, const float* w2,
                 float* u, float* t, float* e, float* weights, float* attended, hipStream_t s, int u_mode,
                 const void* right16, const void* w1_16) {
  GH_REQUIRE(heads >= 1 && heads <= 8, "concat_att: heads=%d not in [1,8]", heads);
  GH_REQUIRE(b > 0 && l > 0 && dr > 0 && ha > 0, "concat_att_fwd: bad sizes");
  GH_REQUIRE(u_mode >= 0 && u_mode <= 2, "concat_att_fwd: u_mode %d not in {0,1,2}", u_mode);
  GH_REQUIRE((goff == nullptr) == (rowg == nullptr), "concat_att_fwd: goff and rowg come together");
  if (!goff) m_real = b * l;
  GH_REQUIRE(m_real >= 0 && m_real <= b * l, "concat_att_fwd: node-compact rows %d do not fit b*l=%d", m_real, b * l);
  if (!rowu && u_mode != 1) nl = b;      // (the u-only call names its row count itself)
  const int M = m_real;
  const bool one_block = ha <= Batch(false, M, s).bn;      // column block of the tile configuration this launch will use
  GH_REQUIRE(one_block || (ha % 4 == 0 && al16(t) && al16(u) && al16(w2)),
             "concat_att_fwd: attention hidden %d wider than one column block needs float4-shaped rows", ha);
  const int xl_in = (left && xl > 0) ? xl : 0;      // column offset of the right branch inside linear1.weight
  if (u_mode != 2) {
    if (left && xl > 0) {  // u = W1[:, :xl] . left -- once per pair / claim, not per token (two_branches_attention.py:137-140)
      Batch bt(false, nl, s);
      bt.add(gemm_problem(nl, ha, EPI_STORE, u, ha, left, xl, w1, xl + dr, xl));
      bt.flush();
      GH_CHECK_HIP(bt.err);
    } else {
      GH_CHECK_HIP(hipMemsetAsync(u, 0, sizeof(float) * (size_t)nl * ha, s));
    }
  }
  if (u_mode == 1) return 0;
  if (!(left && xl > 0)) xl = 0;
  const int32_t* urow = rowu ? rowu : rowg;
  const float* e_parts = nullptr;
  int n_parts = 0;
  bool right_is16 = false;
  if (M > 0) {  // t = tanh(W1[:, xl:] . right_t + u) ; e = W2 t  (:140-141)
    // bf16 storage mode with the twins at hand (right16 = the producing cell's bf16 output, w1_16 = bf16(linear1.weight)): the
    // product runs on the bf16-storage kernel (v_mfma_f32_16x16x32_bf16 from bf16 LDS images, half the operand bytes) instead of
    // rounding fp32 fragments in registers (MODE 1) -- the same bf16 operand values, fp32 accumulation and fp32 t either way
    const bool use16 = right16 && w1_16 && g_gemm_mode == 1 && M >= 8192 && dr % 8 == 0 && (xl_in + dr) % 8 == 0 && xl_in % 8 == 0 && ha % 8 == 0;
    GH_REQUIRE(use16 || right, "concat_att_fwd: no fp32 right operand and the bf16 twins do not apply");
    right_is16 = use16;
    Batch bt(false, M, s, use16 && ha % 256 == 0);
    Problem p = use16 ? gemm_problem(M, ha, EPI_ATT, t, ha, (const float*)right16, dr,
                                     (const float*)((const unsigned short*)w1_16 + xl_in), xl_in + dr, dr, nullptr, 1)
                      : gemm_problem(M, ha, EPI_ATT, t, ha, right, dr, w1 + xl_in, xl_in + dr, dr);
    p.u = u; p.ldu = ha; p.R = l; p.w2 = w2; p.heads = heads; p.e = e; p.rowg = urow;
    // wide hidden layer (h = 768): a row spans several column blocks.  Activation-sized launches let every block reduce its share
    // of the head scores into a partial buffer of its own (stream workspace; att_softmax_fwd sums the partials in block order);
    // without the workspace (or for few-row launches, whose split-K plan finishes whole rows anyway) the product is stored and the
    // tanh + W2 reduction runs as a row-per-wave pass over it.
    bool blocks = false;
    if (!one_block) {
      const int nb = (ha + bt.bn - 1) / bt.bn;
      const Workspace wsp = workspace_for(s);
      const size_t need = (size_t)nb * (size_t)M * heads * sizeof(float);
      if (M >= 8192 && wsp.p && need <= wsp.bytes && ha % 4 == 0) {
        blocks = true;
        bt.e_block_stride = (long long)M * heads;
        p.e = wsp.p;
        e_parts = wsp.p; n_parts = nb;
      } else {
        p.epi = EPI_STORE;
      }
    }
    bt.add(p);
    bt.flush();
    GH_CHECK_HIP(bt.err);
    if (!one_block && !blocks) {
      FinishArgs F;
      F.n = 1; F.split = 0;
      F.it[0] = FinishItem{t, 0, 1, M, ha, EPI_ATT, 0, ha, t, nullptr, nullptr, nullptr, nullptr, nullptr, u, w2, e, ha, l, heads, urow};
      hipLaunchKernelGGL(nt_finish_kernel, dim3((M + 3) / 4, 1), dim3(256), 0, s, F);
      GH_LAUNCH_CHECK();
    }
  }
  // (bf16 storage mode: the weighted sum reads the bf16 rows the product above read -- the caller need not keep an fp32 copy)
  return launch_att_softmax_fwd(e, mask, right, goff, m_real, b, l, dr, heads, weights, attended, s, e_parts, n_parts,
                                (long long)M * heads, right_is16 ? right16 : nullptr);   // (:142-147)
}

extern "C" int gh_concat_att_fwd(const float* left, const float* right, const float* mask, const int32_t* goff,
                                 const int32_t* rowg, int m_real, int b, int l, int xl,
                                 int dr, int ha, int heads, const float* w1, const float* w2,
                                 float* u, float* t, float* e, float* weights, float* attended,
                                 gh_stream_t stream) {
  return att_fwd_impl(left, b, nullptr, right, mask, goff, rowg, m_real, b, l, xl, dr, ha, heads, w1, w2, u, t, e, weights, attended,
                      (hipStream_t)stream);
}

// Backward.  claim_offsets / nl / du_c (all or none): the left input has nl rows, one per CLAIM, and pair p belongs to the
// claim c with claim_offsets[c] <= p < claim_offsets[c+1]: the per-pair du is summed per claim into du_c [nl][ha] first and
// dleft / the left part of dw1 are computed from (du_c, left) over nl rows.  dleft_accumulate: dleft += .
int gh::att_bwd_impl(const float* left, const float* right, const int32_t* goff, int m_real, int b, int l,
                 int xl, int dr, int ha, int heads, const float* w1t, const float* w2, const float* t, const float* weights,
                 const float* g_att, const float* g_w, float* de, float* dpre, float* du,
                 float* dleft, float* dright, float* dw1, float* dw2,
                 const int32_t* claim_offsets, int nl, float* du_c, int dleft_accumulate, hipStream_t s,
                 const int32_t* rowg, float* dw_tmp, const GateFuse* next, int dleft_late, float* dw2_buf,
                 const void* right16, const void* w1t_16) {
  GH_REQUIRE(heads >= 1 && heads <= 8, "concat_att: heads=%d not in [1,8]", heads);
  if (!goff) m_real = b * l;
  GH_REQUIRE(m_real >= 0 && m_real <= b * l, "concat_att_bwd: node-compact rows %d do not fit b*l=%d", m_real, b * l);
  GH_REQUIRE((claim_offsets == nullptr) == (du_c == nullptr), "concat_att_bwd: claim_offsets and du_c come together");
  const int M = m_real;
  if (!left) xl = 0;
  if (!claim_offsets) nl = b;
  const float* dul = claim_offsets ? du_c : du;      // the left branch's pre-activation gradient, one row per left row
  const int ldw = xl + dr;
  // Two-phase use (the caller may put the weight gradient of linear1 on another stream): dw1 == NULL skips its two
  // launches; dright == NULL is the complementary "dw1 only" call on buffers (dpre, du) a first call has filled.
  const bool weights_only = (dright == nullptr);
  GH_REQUIRE(!weights_only || dw1, "concat_att_bwd: dright == NULL asks for the dw1-only phase, which needs dw1");
  // bf16 storage mode with the twins at hand: dpre is produced as bf16 (the products below round it to bf16 anyway) and the
  // dright / dW1 products run on the bf16-storage kernels -- same operand values as the in-register rounding of MODE 1
  const bool use16 = right16 && w1t_16 && g_gemm_mode == 1 && m_real >= 8192 && dr % 8 == 0 && ha % 8 == 0 && xl % 8 == 0;
  void* const dpre16 = use16 ? (void*)dpre : nullptr;
  float* dw2_part = nullptr;
  if (!weights_only) {
  int dw_written = 0;
  if (int e = launch_att_softmax_bwd(right, weights, g_att, g_w, goff, m_real, b, l, dr, heads, de, dright, s, rowg, dw_tmp, &dw_written,
                                     use16 ? right16 : nullptr)) return e;
  // dW2 = de^T t rides along with the dpre pass (per-pair partials in the workspace, one reduce)
  const size_t dw2_bytes = (size_t)b * heads * ha * sizeof(float);
  const Workspace wsp = workspace_for(s);
  const bool late = dleft_late && dw2_buf && ha % 4 == 0;      // partials in the caller's buffer, reduced by the second call
  dw2_part = late ? dw2_buf : ((wsp.p && dw2_bytes <= wsp.bytes && ha % 4 == 0) ? wsp.p : nullptr);
  if (int e = launch_att_dpre(de, w2, t, goff, m_real, b, l, ha, heads, dpre, du, dw2_part, s, dw_written ? dw_tmp : nullptr,
                              dw_written ? weights : nullptr, dw_written ? de : nullptr, dpre16,
                              (long long)m_real * heads, dw_written > 1 ? dw_written : 1)) return e;
  if (dw2_part && !late) {
    ReduceArgs R;
    R.n = 1;
    R.it[0] = ReduceItem{dw2_part, dw2, heads, ha, ha, b, (long long)heads * ha};
    hipLaunchKernelGGL(reduce_partials_wave_kernel, dim3((heads * (ha / 4) + 3) / 4, 1), dim3(256), 0, s, R);
    GH_LAUNCH_CHECK();
  }
  if (claim_offsets && xl > 0 && !dleft_late)
    if (int e = gh_seg_sum(du, claim_offsets, du_c, nl, ha, (gh_stream_t)s)) return e;
  const float* w1t_r16 = use16 ? (const float*)((const unsigned short*)w1t_16 + (size_t)xl * ha) : nullptr;
  GH_REQUIRE(!next || !next->bf16 || use16, "concat_att_bwd: a bf16 gate head needs the bf16 twins of right and w1t");
  if (M > 0 && next) {  // dright (= the gradient of the cell that produced `right`) is consumed by that cell's gate head only:
    Batch bt(false, M, s, use16 && dr % 256 == 0, 6, dr);      // g = softmax part (in dright) + dpre W1[:, xl:] goes straight into dhp / dzp / dxp
    Problem p = use16 ? gemm_problem(M, dr, EPI_GATE_PRE, next->dhp, dr, (const float*)dpre16, ha, w1t_r16, ha, ha, nullptr, 1)
                      : gemm_problem(M, dr, EPI_GATE_PRE, next->dhp, dr, dpre, ha, w1t + (size_t)xl * ha, ha, ha);
    p.gin = dright; p.in0 = next->z; p.in1 = next->hh; p.in2 = next->xp; p.out1 = next->dzp; p.out2 = next->dxp;
    p.io = next->bf16 ? (1 | 2 | 4 | 8 | 16 | 32) : 0;
    bt.add(p);
    bt.flush();
    GH_CHECK_HIP(bt.err);
  } else if (M > 0) {  // dright += dpre W1[:, xl:]
    Batch bt(false, M, s, use16 && dr % 256 == 0);
    Problem p = use16 ? gemm_problem(M, dr, EPI_STORE, dright, dr, (const float*)dpre16, ha, w1t_r16, ha, ha, nullptr, 1)
                      : gemm_problem(M, dr, EPI_STORE, dright, dr, dpre, ha, w1t + (size_t)xl * ha, ha, ha);
    p.accumulate = 1;
    bt.add(p);
    bt.flush();
    GH_CHECK_HIP(bt.err);
  }
  if (xl > 0 && dleft && !dleft_late) {  // dleft = du W1[:, :xl]
    Batch bt(false, nl, s);
    Problem p = gemm_problem(nl, xl, EPI_STORE, dleft, xl, dul, ha, w1t, ha, ha);
    p.accumulate = dleft_accumulate ? 1 : 0;
    bt.add(p);
    bt.flush();
    GH_CHECK_HIP(bt.err);
  }
  if (!dw2_part && M > 0) {  // no workspace: `heads` rows are not float4-shaped, so this one takes the generic kernel
    Batch bt(true, M, s);
    bt.add(tn_problem(heads, ha, dw2, ha, de, heads, t, ha, M));
    bt.flush();
    GH_CHECK_HIP(bt.err);
  }
  }
  if (weights_only && dleft_late) {
    if (dw2_buf && ha % 4 == 0) {      // the first call left the per-pair dW2 partials in dw2_buf
      ReduceArgs R;
      R.n = 1;
      R.it[0] = ReduceItem{dw2_buf, dw2, heads, ha, ha, b, (long long)heads * ha};
      hipLaunchKernelGGL(reduce_partials_wave_kernel, dim3((heads * (ha / 4) + 3) / 4, 1), dim3(256), 0, s, R);
      GH_LAUNCH_CHECK();
    }
    if (claim_offsets && xl > 0)
      if (int e = gh_seg_sum(du, claim_offsets, du_c, nl, ha, (gh_stream_t)s)) return e;
  }
  if (weights_only && dleft_late && xl > 0 && dleft) {  // the left gradient, deferred to this (weight-gradient stream) call
    Batch bt(false, nl, s);
    Problem p = gemm_problem(nl, xl, EPI_STORE, dleft, xl, dul, ha, w1t, ha, ha);
    p.accumulate = dleft_accumulate ? 1 : 0;
    bt.add(p);
    bt.flush();
    GH_CHECK_HIP(bt.err);
  }
  if (!dw1) return 0;
  if (M > 0) {
    Batch bt(true, M, s);
    if (use16) bt.add(tn_problem(ha, dr, dw1 + xl, ldw, (const float*)dpre16, ha, (const float*)right16, dr, M, nullptr, 1));
    else bt.add(tn_problem(ha, dr, dw1 + xl, ldw, dpre, ha, right, dr, M));
    bt.flush();
    GH_CHECK_HIP(bt.err);
  }
  if (xl > 0) {
    Batch bt(true, nl, s);
    bt.add(tn_problem(ha, xl, dw1, ldw, dul, ha, left, xl, nl));
    bt.flush();
    GH_CHECK_HIP(bt.err);
  }
  return 0;
}

extern "C" int gh_concat_att_bwd(const float* left, const float* right, const int32_t* goff, int m_real, int b, int l,
                                 int xl, int dr, int ha, int heads, const float* w1t, const float* w2, const float* t, const float* weights,
                                 const float* g_att, const float* g_w, float* de, float* dpre, float* du,
                                 float* dleft, float* dright, float* dw1, float* dw2, gh_stream_t stream) {
  return att_bwd_impl(left, right, goff, m_real, b, l, xl, dr, ha, heads, w1t, w2, t, weights, g_att, g_w, de, dpre, du, dleft, dright,
                      dw1, dw2, nullptr, b, nullptr, 0, (hipStream_t)stream, nullptr, nullptr, nullptr);
}

// y[m][n] = [x0 | x1] . W^T + bias with W [n][k0 + k1] as stored: the head's first layer on the concatenation
// [claim vector | attended evidences] (graph_based_semantic_structure.py:251-267) without materialising the concatenation.
// w_out / b_out / y_out / n_out (optional): a second linear layer of n_out <= 8 columns on top (the head's out[1], 300 -> 2 classes):
// y_out = y . w_out^T + b_out rides in the split-K finish kernel when the product takes that plan, else one extra launch.
int gh::linear2_fwd(const float* x0, int k0, const float* x1, int k1, const float* w, const float* bias, float* y, int m, int n, hipStream_t s,
                    const float* w_out, const float* b_out, float* y_out, int n_out) {
  Batch b(false, m, s);
  Problem p = gemm_problem(m, n, EPI_STORE, y, n, x0, k0, w, k0 + k1, k0);
  if (k1 > 0) add_seg(p, x1, k1, w + k0, k0 + k1, k1);
  p.bias = bias;
  const bool two = w_out && y_out && n_out >= 1;
  const bool fuse = two && n_out <= 8 && n % 4 == 0 && n <= b.bn && al16(w_out);
  if (fuse) { p.w2 = w_out; p.e = y_out; p.heads = n_out; p.in0 = b_out; }
  b.add(p);
  b.flush();
  GH_CHECK_HIP(b.err);
  if (two && !(fuse && b.split_done)) return launch_tiny_linear_fwd(y, w_out, b_out, y_out, m, n, n_out, s);
  return 0;
}
// dx0 [m][k0] (+)= g Wt[:k0] ; dx1 [m][k1] = g Wt[k0:] (wt = W^T [k0+k1][n]) ; dw [n][k0+k1] += g^T [x0 | x1] ; db += colsum(g).
// dx_only / dw_only split the call in two phases (weight gradients on another stream).
int gh::linear2_bwd(const float* x0, int k0, const float* x1, int k1, const float* wt, const float* g, int m, int n,
                float* dx0, int dx0_accumulate, float* dx1, float* dw, float* db, hipStream_t s) {
  if (dx0 || dx1) {
    Batch b(false, m, s);
    if (dx0) { Problem p = gemm_problem(m, k0, EPI_STORE, dx0, k0, g, n, wt, n, n); p.accumulate = dx0_accumulate ? 1 : 0; b.add(p); }
    if (dx1 && k1 > 0) b.add(gemm_problem(m, k1, EPI_STORE, dx1, k1, g, n, wt + (size_t)k0 * n, n, n));
    b.flush();
    GH_CHECK_HIP(b.err);
  }
  if (dw) {
    Batch b(true, m, s);
    b.add(tn_problem(n, k0, dw, k0 + k1, g, n, x0, k0, m));
    if (k1 > 0) b.add(tn_problem(n, k1, dw + k0, k0 + k1, g, n, x1, k1, m));
    b.flush();
    GH_CHECK_HIP(b.err);
  }
  if (db) return launch_colsum(g, db, m, n, s);
  return 0;
}

extern "C" int gh_set_gemm_mode(int mode) {
  GH_REQUIRE(mode >= 0 && mode <= 3, "set_gemm_mode: %d is not 0 (fp32), 1 (bf16 operands in the big NT/NN GEMMs), 2 (fp32x3) or 3 (fp32x3, pre-split weights)", mode);
  g_gemm_mode = mode;
  return 0;
}

extern "C" int gh_weights_changed(void) {
  std::lock_guard<std::mutex> lk(g_x3_mu);
  ++g_x3_epoch;
  return 0;
}

extern "C" int gh_fp32x3_refresh(gh_stream_t stream) {
  std::lock_guard<std::mutex> lk(g_x3_mu);
  if (g_x3.empty()) return 0;
  std::vector<X3Item> items;
  items.reserve(g_x3.size());
  for (auto& kv : g_x3) {
    items.push_back(X3Item{(const float*)kv.first.b, kv.second.img, kv.first.N, kv.first.ldb, kv.first.K, kv.second.pitch});
    kv.second.epoch = g_x3_epoch;
  }
  x3p_split(items.data(), (int)items.size(), (hipStream_t)stream);
  GH_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gh_fp32x3_clear(void) {
  std::lock_guard<std::mutex> lk(g_x3_mu);
  if (g_x3.empty()) return 0;
  GH_CHECK_HIP(hipDeviceSynchronize());      // a launch on any stream may still read an image; the buffers go back to the pool
  for (auto& kv : g_x3) g_x3_pool.emplace((size_t)kv.first.N * kv.second.pitch + 256, kv.second.img);
  g_x3.clear();
  if (g_x3_pool.size() > 2048) {
    for (auto& kv : g_x3_pool) (void)hipFree(kv.second);
    g_x3_pool.clear();
  }
  return 0;
}

extern "C" int gh_gemm_path_counters(int64_t* out_host, int reset) {
  if (out_host)
    for (int i = 0; i < 3; ++i) out_host[i] = g_path_counts[i];
  if (reset) g_path_counts[0] = g_path_counts[1] = g_path_counts[2] = 0;
  return 0;
}

extern "C" int gh_set_workspace(void* ptr, int64_t bytes) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  g_ws_default = make_workspace(ptr, bytes);
  return 0;
}

extern "C" int gh_set_stream_workspace(gh_stream_t stream, void* ptr, int64_t bytes) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  const WsKey key{current_device(), (hipStream_t)stream};      // the CURRENT device's `stream`
  if (!ptr) g_ws_stream.erase(key);
  else g_ws_stream[key] = make_workspace(ptr, bytes);
  return 0;
}

extern "C" int gh_linear_fwd(const float* x, const float* w, const float* bias, float* y, int m, int k, int n,
                             gh_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  GH_REQUIRE(m > 0 && k > 0 && n > 0, "linear_fwd: bad sizes");
  if (n <= 8) return launch_tiny_linear_fwd(x, w, bias, y, m, k, n, s);      // the 2-class head: one wave per row
  Batch b(false, m, s);
  Problem p = gemm_problem(m, n, EPI_STORE, y, n, x, k, w, k, k);
  p.bias = bias;
  b.add(p);
  b.flush();
  GH_CHECK_HIP(b.err);
  return 0;
}

#ifdef GH_MEASURE
extern "C" int gh_debug_nt_phases(unsigned long long* out, int reset) {     // out: [16][4]
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nt_phase), 64 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_nt_phase)) != hipSuccess || hipMemset(p, 0, 64 * sizeof(unsigned long long)) != hipSuccess) return 1; }
  return 0;
}
#endif

extern "C" int gh_linear_wgrad_bf16(const void* g16, int ldg, const void* x16, int ldx, int m, int n, int k,
                                    float* dw, int lddw, float* db, gh_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  GH_REQUIRE(m > 0 && k > 0 && n > 0 && g16 && x16 && dw, "linear_wgrad_bf16: bad arguments");
  GH_REQUIRE(n % 8 == 0 && k % 8 == 0 && ldg % 8 == 0 && ldx % 8 == 0 && ldg >= n && ldx >= k && lddw >= k && lddw % 4 == 0,
             "linear_wgrad_bf16: widths and leading dimensions must be multiples of 8 (lddw: of 4)");
  GH_REQUIRE(((reinterpret_cast<uintptr_t>(g16) | reinterpret_cast<uintptr_t>(x16) | reinterpret_cast<uintptr_t>(dw)) & 15) == 0,
             "linear_wgrad_bf16: operands must be 16-byte aligned");
  Batch b(true, m, s);
  GH_REQUIRE(b.g_ws != nullptr, "linear_wgrad_bf16: needs the split-K workspace (gh_set_workspace)");
  b.add(tn_problem(n, k, dw, lddw, (const float*)g16, ldg, (const float*)x16, ldx, m, nullptr, 1));
  if (db) b.want_colsum(db, nullptr);
  b.flush();
  GH_CHECK_HIP(b.err);
  GH_REQUIRE(!db || (b.colsum_fused && !b.colsum_missed), "linear_wgrad_bf16: the bias gradient did not fit the split-K workspace");
  return 0;
}

extern "C" int gh_linear_bwd(const float* x, const float* wt, const float* w, const float* g, int m, int k, int n, float* dx,
                             float* dw, float* db, gh_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  GH_REQUIRE(m > 0 && k > 0 && n > 0, "linear_bwd: bad sizes");
  if (n <= 8 && w) return launch_tiny_linear_bwd(x, w, g, dx, dw, db, m, k, n, s);
  if (dx) {
    Batch b(false, m, s);
    b.add(gemm_problem(m, k, EPI_STORE, dx, k, g, n, wt, n, n));
    b.flush();
    GH_CHECK_HIP(b.err);
  }
  if (dw) {
    Batch b(true, m, s);
    b.add(tn_problem(n, k, dw, k, g, n, x, k, m));
    b.flush();
    GH_CHECK_HIP(b.err);
  }
  if (db) return launch_colsum(g, db, m, n, s);
  return 0;
}
