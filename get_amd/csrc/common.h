// Shared host-side helpers of libget_hip.so (error reporting, internal launch prototypes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace gh {

void set_error(const char* fmt, ...);

#define GH_CHECK_HIP(expr)                                                                  \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      gh::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));   \
      return 1;                                                                             \
    }                                                                                       \
  } while (0)

#define GH_REQUIRE(cond, ...)                     \
  do {                                            \
    if (!(cond)) {                                \
      gh::set_error(__VA_ARGS__);                 \
      return 2;                                   \
    }                                             \
  } while (0)

#define GH_LAUNCH_CHECK() GH_CHECK_HIP(hipGetLastError())

inline int words_for(int r) { return (r + 63) / 64; }

// Measurement switches (environment variables GH_*: kernel-variant A/B runs, timing-only modes that skip the K loop or
// the epilogue) exist ONLY in the tool build (`make measure` -> -DGH_MEASURE, lib/libget_hip_measure.so).  The shipped
// library never reads the environment: a stray GH_DBG cannot turn results silently wrong.
inline int measure_env(const char* name, int dflt) {
#ifdef GH_MEASURE
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}
#ifdef GH_MEASURE
#define GH_DBG_BITS(L) ((L).dbg)
#else
#define GH_DBG_BITS(L) 0
#endif

// Optional per-kernel timing with HIP events on the launch stream (bench.py's live roofline numbers).
// Tags index the rows gh_profile_collect() returns; `work` is the launch's algorithmic flops (GEMMs)
// or algorithmic HBM bytes (streaming kernels).
enum ProfTag : int {
  PROF_GEMM_BIG = 0, PROF_GEMM_BIG_TN, PROF_GEMM_SMALL, PROF_GEMM_SMALL_TN,
  PROF_SPMM, PROF_SCORER_GSL, PROF_GRAPH_BUILD, PROF_ATT_SOFTMAX_FWD, PROF_ATT_SOFTMAX_BWD, PROF_ATT_DPRE,
  PROF_GATE_BWD_PRE, PROF_COLSUM, PROF_ADAM,
  PROF_FEW_ROWS,      // claim-side / evidence-level launches of the streaming kernels (a few MB each: launch-latency-bound)
  PROF_NTAGS
};
// streaming-kernel launches over fewer than this many graphs / pairs are booked under PROF_FEW_ROWS
constexpr int PROF_FEW_GROUPS = 256;
unsigned int* clamp_counter();      // four device counters of clamped out-of-range inputs on the current device (gh_clamp_events); NULL = allocation failed
bool prof_enabled();
void prof_begin(hipStream_t s, int tag);   // events are recorded only for tags selected by gh_profile_select
void prof_end(int tag, double work, hipStream_t s);

// gh_scorer_gsl with the projection arriving as score_parts partial tensors, score_stride floats apart (wide cell outputs)
int scorer_gsl_impl(const uint64_t* bits, const float* dinv, const float* vals, const int32_t* goff, int pads_collapsed, const float* feat,
                    const float* score_x, int score_parts, long long score_stride, const float* w_p, const float* gate, int n, int r, int h,
                    int k, float* score, uint64_t* keep, float drop_p, uint32_t drop_seed, hipStream_t stream);
int launch_ragged_plan(const int32_t* n_nodes, const int32_t* node_ids, int n, int r, int32_t* goff, int32_t* rowg, int32_t* src,
                       int32_t* cids, float* maskf, const int64_t* slot, int32_t* document, hipStream_t s, bool* scattered);   // *scattered: the document scatter rode along
int launch_graph_build2(const int32_t* ta, const int32_t* la, int na, int ra, int32_t* ida, int32_t* nna, uint64_t* ba, float* da,
                        const int32_t* tb, const int32_t* lb, int nb, int rb, int32_t* idb, int32_t* nnb, uint64_t* bb, float* db,
                        int window, hipStream_t s);
struct ZeroFill { float* p0; long long n0; float* p1; long long n1; };      // n in floats, multiples of 4, 16-byte aligned pointers
// internal launchers implemented in graph_ops.hip / misc_ops.hip, used by the fused entry points
// goff != NULL: node-compact layout (include/get_hip.h) -- graph g owns rows [goff[g], goff[g+1]); m_real = goff[n]
int launch_spmm(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep, const int32_t* goff,
                int m_real, const float* x, float* y, int n, int r, int h, int transpose, int accumulate, hipStream_t s,
                int bf16 = 0,       // bf16: x and y hold bf16
                const ZeroFill* zf = nullptr, bool* zf_done = nullptr);
// zf: up to two float ranges the launch zero-fills on its way (the cell forward's padding rows of `a` and the scorer's partial
// dot products: two memset dispatches per step otherwise); *zf_done says whether the kernel variant that ran supports it
int launch_gate_bwd_pre(const float* g, const float* z, const float* hh, const float* xp, float* dhp, float* dzp,
                        float* dxp, size_t count, hipStream_t s, int bf16 = 0);   // bf16: everything but g holds bf16
int launch_colsum3(const float* a, const float* b, const float* c, float* oa, float* ob, float* oc, int m, int h,
                   hipStream_t s, float* oa2 = nullptr, float* ob2 = nullptr, float* oc2 = nullptr);
int launch_colsum(const float* a, float* oa, int m, int h, hipStream_t s);
// Scratch registered by the caller: per stream (gh_set_stream_workspace) or the process default (gh_set_workspace).
// `p` serves the split-K partial tiles, `cs` (the last 1/16) the column-sum partials.
struct Workspace { float* p; size_t bytes; float* cs; size_t cs_bytes; };
Workspace workspace_for(hipStream_t s);
int launch_gather_rows(const float* table, const int32_t* ids, float* dst, int m, int d, hipStream_t s,
                       float drop_p = 0.f, unsigned drop_seed = 0, int bf16 = 0);   // bf16: table and dst hold bf16
int launch_tiny_linear_fwd(const float* x, const float* w, const float* bias, float* y, int m, int k, int n, hipStream_t s);
// head backward in one launch (misc_ops.hip head_bwd_kernel): d_y0 = g_phi w1; [dx0 | dx1] = d_y0 w0 (w0 as stored, [H][E]); dw1 / db1 +=
int launch_head_bwd(const float* g_phi, const float* y0, const float* w1, const float* w0, int B, int H, int C, int Xl, int E,
                    float* d_y0, float* dw1, float* db1, float* dx0, int dx0_accumulate, float* dx1, hipStream_t s);
size_t head_bwd_lds(int B, int H, int C, int E, const float* w0);      // 0: the sizes do not fit its LDS staging
int launch_tiny_linear_bwd(const float* x, const float* w, const float* g, float* dx, float* dw, float* db, int m, int k, int n,
                           hipStream_t s);
int launch_att_softmax_fwd(float* e, const float* mask, const float* right, const int32_t* goff, int m_real,
                           int b, int l, int dr, int heads, float* weights, float* attended, hipStream_t s,
                           const float* e_parts = nullptr, int n_parts = 0, long long part_stride = 0, const void* right16 = nullptr);
// (e_parts: n_parts partial score tensors [rows][heads], part_stride floats apart -- one per column block of a wide hidden
//  layer; the kernel sums them in block order and writes the sum to e)
int launch_att_softmax_bwd(const float* right, const float* weights, const float* g_att, const float* g_w,
                           const int32_t* goff, int m_real, int b, int l, int dr, int heads, float* de, float* dright,
                           hipStream_t s, const int32_t* rowg = nullptr, float* dw_tmp = nullptr, int* dw_written = nullptr,
                           const void* right16 = nullptr);
// (right16: the right operand as bf16 rows -- bf16 storage mode; `right` may then be NULL on the row-balanced / forward kernels)
int gemm_mode();      // gh_set_gemm_mode's current value
// (rowg + dw_tmp [ceil(dr / 512)][rows][heads] + dw_written: many-pair launches may take the row-balanced kernel, which leaves the raw
//  dw in dw_tmp instead of de -- *dw_written = the number of column ranges (<= 16) rows wider than 512 floats were split into, whose
//  partial dw sit rows * heads floats apart -- and launch_att_dpre(dw_in = dw_tmp, weights, de_out = de, .., stride, ranges) finishes de)
int launch_att_dpre(const float* de, const float* w2, const float* t, const int32_t* goff, int m_real, int b, int l,
                    int ha, int heads, float* dpre, float* du, float* dw2_part, hipStream_t s, const float* dw_in = nullptr,
                    const float* weights = nullptr, float* de_out = nullptr, void* dpre16 = nullptr,       // dpre16: dpre as bf16 there instead
                    long long dw_range_stride = 0, int dw_ranges = 1);      // dw_ranges > 1: dw_in = that many column-range partials of att_rows_bwd (rows wider than 512 floats)

// Where the gradient w.r.t. a GGNN cell's output goes when the GEMM that produces it applies that cell's gate head in its
// epilogue (EPI_GATE_PRE): the cell's saved z / hh / xp and its dhp / dzp / dxp scratch (all [rows][h] fp32).
struct GateFuse { const float* z; const float* hh; const float* xp; float* dhp; float* dzp; float* dxp; int bf16; };      // bf16: all six hold bf16 (bf16 storage pipeline)

// fused building blocks shared by the per-module entry points (gemm_ops.hip) and the composite model entry points
// (model_ops.hip); argument meaning as the gh_* functions of the same name in include/get_hip.h
int cell_fwd_impl(int bf, float* out32, const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                  const int32_t* goff, int m_real, int m_rows, const float* x, const int32_t* ids, int n, int r, int din, int h,
                  const float* w_p, const float* w_z0, const float* w_z1, const float* w_r0, const float* w_r1, const float* w_h0,
                  const float* w_h1, const float* b_z0, const float* b_z1, const float* b_r0, const float* b_r1, const float* b_h0,
                  const float* b_h1, float* xp, float* a, float* z, float* rr, float* rx, float* hh, float* out, float drop_p,
                  uint32_t drop_seed, const float* score_w, float* score_x, float score_drop_p, uint32_t score_drop_seed,
                  void* stream, int pad_out_dead = 0,       // pad_out_dead: nobody reads `out` of the padding rows (fused scorer only)
                  int* score_parts = nullptr,    // non-NULL: score_x holds ceil(h / block) x m_rows floats and a wide row's projection may
                                                 // arrive as per-block partials; *score_parts = how many (1 = plain)
                  float* xdrop = nullptr);       // bf16 storage, drop_p > 0: scratch [m_rows][din] bf16 that receives the masked (gathered) operand
                                                 // rows of the projection; hand the same buffer to cell_bwd_impl
int cell_bwd_impl(int bf, const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep, const int32_t* goff,
                  int m_real, const float* x, const int32_t* ids, int n, int r, int din, int h, const float* wt_p,
                  const float* wt_z0, const float* wt_z1, const float* wt_r0, const float* wt_r1, const float* wt_h0,
                  const float* wt_h1, const float* xp, const float* a, const float* z, const float* rr, const float* rx,
                  const float* hh, const float* g, float* dhp, float* dzp, float* drp, float* dxp, float* da, float* dx,
                  float* dw_p, float* dw_z0, float* dw_z1, float* dw_r0, float* dw_r1, float* dw_h0, float* dw_h1, float* db_z,
                  float* db_r, float* db_h, float* db_z1, float* db_r1, float* db_h1, float drop_p, uint32_t drop_seed, void* stream,
                  void* wstream, hipEvent_t ev_l1, hipEvent_t ev_agg,       // wstream: weight-gradient stream (NULL = stream)
                  int pre_done, const GateFuse* next,    // pre_done: dhp / dzp / dxp already hold the gate head (skip gate_bwd_pre);
                                                         // next: write the dX product into the previous cell's gate head instead of dx
                  const float* xdrop = nullptr);         // the forward's masked operand rows (cell_fwd_impl xdrop), or NULL
int att_fwd_impl(const float* left, int nl, const int32_t* rowu, const float* right, const float* mask, const int32_t* goff,
                 const int32_t* rowg, int m_real, int b, int l, int xl, int dr, int ha, int heads, const float* w1, const float* w2,
                 float* u, float* t, float* e, float* weights, float* attended, hipStream_t s, int u_mode = 0,
                 const void* right16 = nullptr, const void* w1_16 = nullptr);      // bf16 twins of `right` and of w1 (bf16 storage mode)
int att_bwd_impl(const float* left, const float* right, const int32_t* goff, int m_real, int b, int l, int xl, int dr, int ha,
                 int heads, const float* w1t, const float* w2, const float* t, const float* weights, const float* g_att,
                 const float* g_w, float* de, float* dpre, float* du, float* dleft, float* dright, float* dw1, float* dw2,
                 const int32_t* claim_offsets, int nl, float* du_c, int dleft_accumulate, hipStream_t s,
                 const int32_t* rowg = nullptr, float* dw_tmp = nullptr, const GateFuse* next = nullptr, int dleft_late = 0,
                 float* dw2_buf = nullptr, const void* right16 = nullptr, const void* w1t_16 = nullptr);
// right16 / w1t_16 (bf16 storage mode, both calls of a two-phase backward alike): bf16 twins of `right` and of w1t; dpre is then
// produced as bf16 (in the first half of the caller's dpre buffer) and the dright / dW1 products run on the bf16-storage kernels
// att_fwd_impl u_mode: 0 = the whole layer; 1 = ONLY the left projection u (the caller runs it early, e.g. on the claim branch's
//   stream); 2 = u is already there.  att_bwd_impl dleft_late: the left gradient's GEMM runs in the dw1-only call (dright == NULL)
//   instead of the first one, i.e. on the weight-gradient stream, off the critical path; with dw2_buf ([b][heads][ha] floats, the same
//   pointer in both calls) the per-pair dW2 partials go there instead of the stream workspace and their reduction, like the
//   per-claim sum of du, moves into the second call as well.
int linear2_fwd(const float* x0, int k0, const float* x1, int k1, const float* w, const float* bias, float* y, int m, int n, hipStream_t s,
                const float* w_out = nullptr, const float* b_out = nullptr, float* y_out = nullptr, int n_out = 0);
int linear2_bwd(const float* x0, int k0, const float* x1, int k1, const float* wt, const float* g, int m, int n, float* dx0,
                int dx0_accumulate, float* dx1, float* dw, float* db, hipStream_t s);

}  // namespace gh
