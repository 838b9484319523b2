/*
 * get_hip.h -- C-ABI of libget_hip.so, the MI355X (gfx950) implementation of the
 * CRIPAC-DIG/GET hot path: sliding-window word-graph build, gated graph cells with
 * graph-structure-learning (GSL) top-k refinement, and the word-/evidence-level
 * multi-head "concat" attention.
 *
 * The reference has no FFI boundary of its own (it is pure Python on ATen); its
 * boundary is the nn.Module API listed in SURVEY.md section 8(b).  Every entry
 * point below names the reference function whose device work it replaces
 * (paths relative to the upstream tree).  get_amd/ mirrors the reference's
 * module classes on top of these.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; all float data is
 *     fp32, dense, row-major; ids are int32; adjacency bit rows are uint64 words
 *   - nothing allocates: outputs and scratch are caller-provided
 *   - stream-ordered; `stream` is a hipStream_t passed as void*.  Calls carry all their state in their arguments
 *     except three pieces of process configuration: the split-K scratch (registered per stream with
 *     gh_set_stream_workspace, or one default buffer with gh_set_workspace -- concurrent streams that both run
 *     backward passes need one buffer each), the GEMM arithmetic mode (gh_set_gemm_mode, set before launching work)
 *     and the measurement hook (gh_profile_*, single stream by design)
 *   - returns 0 on success, non-zero on error (message via gh_last_error());
 *     never aborts the process
 *
 * Packed adjacency of one graph with R (padded) nodes, W = ceil(R/64):
 *   bits[R][W]  uint64  bit j of row i set  <=>  A[i][j] != 0
 *   dinv[R]     float   deg(i)^-1/2 of the binary graph (0 for padding rows);
 *                       edge weight = dinv[i]*dinv[j]       ("normalised" mode)
 *   vals[R][R]  float   optional dense edge weights; when non-NULL they are used
 *                       instead of dinv ("weighted" mode: any dense adjacency
 *                       the reference API hands over)
 *   keep[W]     uint64  optional GSL keep-set; refined edge (i,j) exists iff
 *                       bit(i,j) && (keep(i) || keep(j))    (wrapper.py:221-225)
 *
 * Node-compact row layout (optional; `goff` != NULL selects it, NULL keeps the reference's padded layout).
 * The reference pads every evidence graph to R nodes and runs all of them through the GGNN cells
 * (graph_based_semantic_structure.py:99-107); a padding node has no edges and is masked out of the attention
 * (:180), so it never influences a real node's output or any gradient -- its only observable effect is that its
 * score competes in GSL's top-k (wrapper.py:216-219).  In the compact layout the n graphs' REAL nodes are stored
 * back to back and all padding nodes after them:
 *   goff[n+1]   int32   goff[g] = first feature row of graph g's real nodes; m_real = goff[n]
 *   node j <  n_g of graph g  ->  feature row goff[g] + j
 *   node j >= n_g of graph g  ->  feature row m_real + (g*R - goff[g]) + (j - n_g)         (m_tot = n*R rows in all)
 * Kernels that only matter for real nodes (second cell, attention, every backward) then run on the first m_real
 * rows; the first cell's forward and the scorer still see all m_tot rows.  bits/dinv/vals/keep, scores and ids keep
 * their padded [g][R] indexing.  `m_real` is passed by value, so the host must know it (sum of the node counts).
 */
#ifndef GET_HIP_H
#define GET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gh_stream_t;

#define GH_ABI_VERSION 10

int gh_abi_version(void);
/* Thread-local message of the last failing call on this thread (never NULL). */
const char* gh_last_error(void);

/* ---- a1  graph build: interactions.py:334-351 convert_text + :11-18 _laplacian_normalize ----
 * tokens[n_texts][fixed_length] raw token sequence (post-padded), lengths[n_texts].
 * Out: node_ids[n_texts][fixed_length] de-duplicated ids in first-occurrence order (0 padded),
 *      n_nodes[n_texts], bits[n_texts][R][W], dinv[n_texts][R]   (R = fixed_length). */
int gh_graph_build(const int32_t* tokens, const int32_t* lengths, int n_texts, int fixed_length, int window,
                   int32_t* node_ids, int32_t* n_nodes, uint64_t* bits, float* dinv, gh_stream_t stream);

/* Dense adjacency handed over by the reference API ((N,R,R) float64 from handlers/mz_sampler.py:146,
 * cast by `.float()` at graph_based_semantic_structure.py:99,149) -> packed bits + fp32 values.  The bit pattern is the
 * union of A's and A^T's non-zero patterns (entries present on one side only carry the value 0), so that the
 * transposed aggregation of the backward pass sees every entry of an adjacency whose pattern is not symmetric. */
int gh_adj_pack_f64(const double* adj, int n, int r, uint64_t* bits, float* vals, gh_stream_t stream);
int gh_adj_pack_f32(const float* adj, int n, int r, uint64_t* bits, float* vals, gh_stream_t stream);

/* De-padding of what the reference fitter holds before its per-claim loop
 * (Fitting/FittingFC/char_man_fitter_query_repr1.py:196-250: `evd_doc_contents` (b, n_max, r) ids and `evd_docs_adj`
 * (b, n_max, r, r) float64, `[:evd_count]` sliced claim by claim with two host syncs each) in one launch:
 * pair p = sum(counts[:c]) + j for every slot j < counts[c] (counts clamped to [0, n_max]) gets its ids narrowed to int32
 * (d_ids [b*n_max][r], rows [0, pairs) written), its adjacency packed as gh_adj_pack_f64 does (bits [b*n_max][r][W],
 * vals [b*n_max][r][r] -- see below) and its number of real nodes (id >= 1) in n_nodes [b*n_max].
 * The adjacency convert_text produces (interactions.py:11-18: D^-1/2 A D^-1/2 of a binary graph) is RECOGNISED: for such a
 * pair only bits and dinv [b*n_max][r] (= 1 / sqrt(row degree), exactly as gh_graph_build writes it) are written -- the
 * kernels' "normalised" mode -- and vals stays untouched, unless force_vals != 0.
 * stats[5] (device, int64): {pairs, real nodes over all pairs, pairs whose ids are not prefix-shaped or whose padding
 * nodes carry edges -- must be 0 for the node-compact layout (gh_ragged_plan) to apply --, pairs whose values are NOT the
 * normalised graph (0: hand bits + dinv to the kernels; > 0: the batch needs the weighted mode -- call again with
 * force_vals = 1 so that vals is complete), reserved}. */
int gh_ref_depad(const int64_t* counts, int b, int n_max, int r, const void* ids, int ids_i64, const double* adj,
                 int32_t* d_ids, uint64_t* bits, float* vals, float* dinv, int32_t* n_nodes, int64_t* stats, int force_vals,
                 gh_stream_t stream);

/* Node-compact layout plan from the node counts of gh_graph_build (device arrays):
 *   goff[n+1] (see above); rowg[n*r] graph of every compact row; src[n*r] padded row index g*r+j of every compact
 *   row (scatter/gather between the two layouts); cids[n*r] = node_ids[src[.]] and maskf[n*r] = (cids >= 1) as float,
 *   the word attention's mask (both NULL ok: not written). */
int gh_ragged_plan(const int32_t* n_nodes, const int32_t* node_ids, int n, int r, int32_t* goff, int32_t* rowg,
                   int32_t* src, int32_t* cids, float* maskf, gh_stream_t stream);

/* ---- aggregation  a = A_hat x : Models/BiDAF/wrapper.py:192 `adj.matmul(x)` ----
 * x,y [n][r][h] (goff == NULL) or node-compact [m_real][h].  transpose: use A^T (backward of the weighted mode);
 * accumulate: y += . */
int gh_spmm(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
            const int32_t* goff, int m_real, const float* x, float* y, int n, int r, int h, int transpose,
            int accumulate, gh_stream_t stream);

/* The same aggregation on the bf16 storage pipeline's activations (what gh_ggnn_cell_fwd_bf16 / _bwd_bf16 run internally, exposed for
 * the parity test and the micro-benchmark): x16, y16 hold bf16, 16-byte aligned rows, h % 8 == 0.  Graphs of r <= 128 nodes run as a
 * dense product per graph on the matrix pipe (the fp32 edge weights split three ways into bf16: products exact to fp32 precision,
 * fp32 accumulation), larger ones on the edge-list kernel (fp32 sums in the fp32 kernel's order); either way ONE rounding to bf16
 * at the store (accumulate: y16 is read, added in fp32, rounded once). */
int gh_spmm_bf16(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                 const int32_t* goff, int m_real, const void* x16, void* y16, int n, int r, int h, int transpose,
                 int accumulate, gh_stream_t stream);

/* ---- weight packing: W[n_out][n_in] -> Wt[n_in][n_out] ----
 * The MFMA GEMMs take both operands contraction-contiguous: forward products x.W^T use the weight exactly as
 * PyTorch stores it, the backward's dX = g.W needs the transposed copy made here (refreshed once per optimiser step). */
int gh_transpose(const float* w, float* wt, int rows, int cols, gh_stream_t stream);
/* All weights of a model in one launch: n matrices, HOST arrays of device pointers and sizes. */
int gh_transpose_batch(int n, const void* const* src_host, void* const* dst_host, const int* rows_host,
                       const int* cols_host, gh_stream_t stream);

/* ---- a2  GGNN cell: Models/BiDAF/wrapper.py:188-208 GGNN.forward ----
 * Input rows are x[m][din] (m = n*r), or emb[ids[m]][din] when ids != NULL (fused
 * embedding gather, graph_based_semantic_structure.py:100,150).
 * w_*: the reference's linear.weight tensors as stored: w_p[h][din]; w_z0,w_z1,w_r0,w_r1,w_h0,w_h1 [h][h].
 * b_?0, b_?1: the two biases of every gate (each [h]); the epilogues add both.
 * Saved for backward (all [m][h]): xp, a, z, r, rx, hh.  out [m][h].
 * drop_p > 0: the cell's input dropout (wrapper.py:185-190) is applied inside the first GEMM's loader with a
 * stateless mask -- element (m,k) kept iff hash(drop_seed, m*din+k) >= drop_p*2^32, scaled by 1/(1-drop_p);
 * pass the same (drop_p, drop_seed) to the backward.  Needs din % 4 == 0 and h % 4 == 0.
 * goff != NULL: node-compact layout; the cell runs on the first m_rows rows (m_real <= m_rows <= n*r; rows beyond
 * m_real are padding nodes: no neighbours).  goff == NULL: m_real/m_rows are ignored, m = n*r.
 * NOTE (node-compact, m_rows > m_real): rows >= m_real of the SAVED tensors r (`rr`) and h~ (`hh`) are UNDEFINED on return --
 * they are backward-only state and the backward never touches a padding row, so the fast epilogues do not store them
 * (xp, a, z, rx and out ARE written for every row < m_rows).  Do not dump or reuse those rows.
 * score_w/score_x (both or neither; needs h % 4 == 0 and h <= 320): the GSL word scorer that consumes this cell's
 * output (wrapper.py:167, GGNN(h->1)) starts with proj(dropout(out)), a [m][h] x [h] product; with score_w[h] =
 * that proj weight the last GEMM's epilogue writes score_x[m] = dropout(out)[m] . score_w while `out` is still in
 * registers (mask = the scorer's own: hash(score_drop_seed, m*h+k) >= score_drop_p*2^32), and gh_scorer_gsl then
 * takes score_x instead of re-reading `out`. */
int gh_ggnn_cell_fwd(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                     const int32_t* goff, int m_real, int m_rows,
                     const float* x, const int32_t* ids, int n, int r, int din, int h,
                     const float* w_p, const float* w_z0, const float* w_z1, const float* w_r0,
                     const float* w_r1, const float* w_h0, const float* w_h1,
                     const float* b_z0, const float* b_z1, const float* b_r0, const float* b_r1,
                     const float* b_h0, const float* b_h1,
                     float* xp, float* a, float* z, float* rr, float* rx, float* hh, float* out,
                     float drop_p, uint32_t drop_seed,
                     const float* score_w, float* score_x, float score_drop_p, uint32_t score_drop_seed,
                     gh_stream_t stream);

/* Backward of the cell.  wt_*: TRANSPOSED weights (gh_transpose of linear.weight): wt_p[din][h], the others [h][h].
 * g [m][h] = dL/dout.  Scratch (all [m][h]): dhp, dzp, drp, dxp, da.
 * Outputs: dx [m][din] (may be NULL: frozen embedding), and ACCUMULATED (+=) into
 * dw_p[h][din], dw_z0..dw_h1 [h][h], db_z[h], db_r[h], db_h[h]  (caller zeroes them, or hands the
 * parameters' own .grad buffers).  b?0 and b?1 share one gradient: db_z1/db_r1/db_h1 (NULL ok) receive
 * the same column sums, so both biases' .grad can be fed without a copy.
 * goff != NULL: node-compact layout, the backward runs on the m_real real-node rows only (padding rows get and give
 * no gradient; dx rows beyond m_real are not written). */
int gh_ggnn_cell_bwd(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                     const int32_t* goff, int m_real,
                     const float* x, const int32_t* ids, int n, int r, int din, int h,
                     const float* wt_p, const float* wt_z0, const float* wt_z1, const float* wt_r0,
                     const float* wt_r1, const float* wt_h0, const float* wt_h1,
                     const float* xp, const float* a, const float* z, const float* rr, const float* rx,
                     const float* hh, const float* g,
                     float* dhp, float* dzp, float* drp, float* dxp, float* da,
                     float* dx, float* dw_p, float* dw_z0, float* dw_z1, float* dw_r0, float* dw_r1,
                     float* dw_h0, float* dw_h1, float* db_z, float* db_r, float* db_h,
                     float* db_z1, float* db_r1, float* db_h1, float drop_p, uint32_t drop_seed, gh_stream_t stream);

/* ---- bf16 STORAGE variant of the cell (BASELINE configs[4]: "h=768 bf16 ... MFMA projections") ----
 * Same computation and argument order as gh_ggnn_cell_fwd / _bwd with these tensors holding bf16 instead of fp32:
 *   forward : x (rows, or the embedding table behind ids), the seven weights, xp a z rr rx hh, out;
 *             out32 [m][h] fp32 additionally receives the cell output for the fp32 consumers (attention, scorer stays fused)
 *   backward: x / table, the seven TRANSPOSED weights, the saved xp..hh, the scratch dhp dzp drp dxp da
 * Biases, g, dx, score_w/score_x and every weight/bias gradient stay fp32; all arithmetic accumulates in fp32
 * (v_mfma_f32_16x16x32_bf16; gate math in fp32 registers).  Needs din % 8 == 0, h % 8 == 0, din <= h, at least 8192 rows
 * and the split-K workspace.  Results follow the fp32 path to bf16 accuracy (tests/test_gpu_model.py states the tolerances). */
int gh_ggnn_cell_fwd_bf16(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                          const int32_t* goff, int m_real, int m_rows,
                          const void* x, const int32_t* ids, int n, int r, int din, int h,
                          const void* w_p, const void* w_z0, const void* w_z1, const void* w_r0,
                          const void* w_r1, const void* w_h0, const void* w_h1,
                          const float* b_z0, const float* b_z1, const float* b_r0, const float* b_r1,
                          const float* b_h0, const float* b_h1,
                          void* xp, void* a, void* z, void* rr, void* rx, void* hh, void* out, float* out32,
                          float drop_p, uint32_t drop_seed,
                          const float* score_w, float* score_x, float score_drop_p, uint32_t score_drop_seed,
                          gh_stream_t stream);
int gh_ggnn_cell_bwd_bf16(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                          const int32_t* goff, int m_real,
                          const void* x, const int32_t* ids, int n, int r, int din, int h,
                          const void* wt_p, const void* wt_z0, const void* wt_z1, const void* wt_r0,
                          const void* wt_r1, const void* wt_h0, const void* wt_h1,
                          const void* xp, const void* a, const void* z, const void* rr, const void* rx,
                          const void* hh, const float* g,
                          void* dhp, void* dzp, void* drp, void* dxp, void* da,
                          float* dx, float* dw_p, float* dw_z0, float* dw_z1, float* dw_r0, float* dw_r1,
                          float* dw_h0, float* dw_h1, float* db_z, float* db_r, float* db_h,
                          float* db_z1, float* db_r1, float* db_h1, float drop_p, uint32_t drop_seed, gh_stream_t stream);

/* ---- a2(300->1) + a3  word scorer + GSL top-k: wrapper.py:167-168, GSL.forward :215-227 ----
 * feat [n][r][h]; w_p[h] = scorer proj.linear.weight; gate[12] = {wz0,bz0,wz1,bz1,wr0,br0,wr1,br1,
 * wh0,bh0,wh1,bh1} (the six 1x1 linears).  k = int(rate * r) computed by the caller.
 * Out: score[n][r], keep[n][W] (bit i set <=> node i among the k best; ties -> lower index).
 * drop_p/drop_seed: the scorer cell's own input dropout in training mode (same stateless mask as above).
 * goff != NULL: feat is node-compact [n*r][h] INCLUDING the padding rows (they compete in the top-k).
 * pads_collapsed (with goff, drop_p == 0 only): without dropout all padding rows of a batch are identical, so feat
 * holds just ONE of them, at row goff[n] (feat is [goff[n] + 1][h]); every padding node scores with that row.
 * Exactly one of feat / score_x is non-NULL: score_x[rows] = the projections proj(dropout(feat)) already produced by
 * gh_ggnn_cell_fwd's epilogue (same row indexing as feat; w_p, h and the dropout arguments are then unused). */
int gh_scorer_gsl(const uint64_t* bits, const float* dinv, const float* vals, const int32_t* goff, int pads_collapsed,
                  const float* feat, const float* score_x, const float* w_p, const float* gate, int n, int r, int h, int k,
                  float* score, uint64_t* keep, float drop_p, uint32_t drop_seed, gh_stream_t stream);
/* GSL alone on given scores (GSL.forward on arbitrary score input). */
int gh_gsl_topk(const float* score, int n, int r, int k, uint64_t* keep, gh_stream_t stream);
/* Dense view of a (refined) packed adjacency, for callers that want GSL.forward's dense result. */
int gh_adj_unpack(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                  int n, int r, float* adj, gh_stream_t stream);

/* ---- a5/a6  concat attention: thirdparty/two_branches_attention.py:121-148 (left != NULL) and
 *      thirdparty/self_attention.py:75-100 (left == NULL) ----
 * left [b][xl] (or NULL, xl = 0), right [b][l][dr], mask [b][l] float (0 = padded).
 * w1 = linear1.weight [ha][xl+dr] as stored; w2 = linear2.weight [heads][ha] (heads <= 8).
 * Saved: u [b][ha] (left branch, hoisted out of the per-token product), t [b*l][ha] (tanh), e [b*l][heads].
 * Out: weights [b][l][heads], attended [b][dr][heads].
 * goff/rowg != NULL (both): right, mask, t, e, weights are node-compact with m_real rows, pair g owning rows
 * [goff[g], goff[g+1]) (at most l of them) and rowg[row] = g; softmax runs over the pair's real rows only. */
int gh_concat_att_fwd(const float* left, const float* right, const float* mask, const int32_t* goff,
                      const int32_t* rowg, int m_real, int b, int l, int xl, int dr,
                      int ha, int heads, const float* w1, const float* w2,
                      float* u, float* t, float* e, float* weights, float* attended, gh_stream_t stream);
/* Backward.  w1t = transpose of linear1.weight: [xl+dr][ha].  g_att [b][dr][heads], g_w [b][l][heads] or NULL.
 * Scratch: de [b*l][heads], dpre [b*l][ha], du [b][ha].
 * Out: dleft [b][xl] (NULL ok), dright [b][l][dr]; ACCUMULATED: dw1 [ha][xl+dr], dw2 [heads][ha].
 * Two-phase use (e.g. the weight gradient of linear1 on another stream): dw1 == NULL leaves its launches out;
 * dright == NULL is the complementary call that computes ONLY dw1 from the dpre / du a first call has filled. */
int gh_concat_att_bwd(const float* left, const float* right, const int32_t* goff, int m_real, int b, int l, int xl,
                      int dr, int ha, int heads,
                      const float* w1t, const float* w2, const float* t, const float* weights,
                      const float* g_att, const float* g_w,
                      float* de, float* dpre, float* du,
                      float* dleft, float* dright, float* dw1, float* dw2, gh_stream_t stream);

/* ---- GEMM arithmetic mode (process-wide, default 0) ----
 * 0: exact fp32 MFMA everywhere -- the mode every parity claim and the headline benchmark are made in.
 * 1: (the host layer additionally routes the evidence cells through gh_ggnn_cell_*_bf16 in this mode) the big-tile GEMMs (cell gates, dX, attention projections with >= 8192 rows, and the split-K weight gradients)
 *    round their operands to bf16 while staging them in LDS and use v_mfma_f32_16x16x16_bf16 with fp32 accumulation;
 *    storage, epilogues, bias gradients and all small GEMMs stay fp32.  For BASELINE configs[4] ("h=768 bf16"); results then
 *    match the fp32 oracle to bf16 accuracy only (~1e-2 relative on logits). */
int gh_set_gemm_mode(int mode);
/* 2 ("fp32x3", opt-in): fp32 storage and results; every product of the activation-sized NT launches is formed on the bf16 MFMA
 *    from 3-way bf16 splits of both operands (six of the nine cross terms; error below one fp32 rounding of the product).
 * 3 ("fp32x3 with pre-split weights", opt-in): as 2, with the weight operand's pieces read from an image the library keeps
 *    per weight view (made on first use).  The caller tells the library when weights were rewritten: gh_weights_changed()
 *    marks every image stale (a stale image is re-made by the launch that meets it), gh_fp32x3_refresh(stream) re-makes all
 *    of them in one launch (call it after the optimiser step), gh_fp32x3_clear() frees them (before weights are freed). */
int gh_weights_changed(void);
int gh_fp32x3_refresh(gh_stream_t stream);
int gh_fp32x3_clear(void);

/* ---- split-K scratch for the weight-gradient GEMMs ----
 * Caller-owned device buffer (stays registered until replaced; NULL unregisters).  With it the
 * K-chunk partial tiles are written with plain stores and reduced by a second kernel; without it
 * (or when it is too small) the chunks add into the output with fp32 atomics.  All work that uses
 * it is ordered on the stream of the call, so one buffer serves one stream at a time. */
int gh_set_workspace(void* ptr, int64_t bytes);
/* The same, for work launched on `stream` of the CURRENT device only (takes precedence over the default buffer; NULL ptr
 * unregisters).  The registry is keyed by (hipGetDevice(), stream): the default stream has handle 0 on every device. */
int gh_set_stream_workspace(gh_stream_t stream, void* ptr, int64_t bytes);

/* ---- plain linear y = x W^T + b (model head, graph_based_semantic_structure.py:69-72) ---- */
int gh_linear_fwd(const float* x, const float* w, const float* bias, float* y, int m, int k, int n,
                  gh_stream_t stream);
/* dx = g W (NULL ok; wt = W^T [k][n]); dw += g^T x; db += colsum(g) (NULL ok).  w = W as stored (NULL ok): layers with
 * n <= 8 outputs (the 2-class head) then run as one row-per-wave kernel instead of three MFMA launches. */
int gh_linear_bwd(const float* x, const float* wt, const float* w, const float* g, int m, int k, int n,
                  float* dx, float* dw, float* db, gh_stream_t stream);

/* Weight gradient of a linear layer of the bf16 storage pipeline (gh_ggnn_cell_bwd_bf16's building block, exposed for callers'
 * own layers and for the exact parity test of the kernel): g16 [m][ldg] and x16 [m][ldx] hold bf16 (n resp. k columns used),
 *     dw[n][k] (fp32, leading dimension lddw) += g^T x,      db[n] (fp32, NULL ok) += colsum(g),
 * products exact, fp32 accumulation, a fixed summation order.  Needs the split-K workspace (gh_set_workspace), 16-byte aligned
 * operands with ldg % 8 == ldx % 8 == 0 and n % 8 == k % 8 == 0.  Outputs whose widths are multiples of 256 over >= 16 384 rows run
 * on the 256 x 256 x 64 ping-pong tile (gemm_tn_pp.hip.h), the rest on the 128 x 320 tile (gemm_tn.hip.h). */
int gh_linear_wgrad_bf16(const void* g16, int ldg, const void* x16, int ldx, int m, int n, int k,
                         float* dw, int lddw, float* db, gh_stream_t stream);

/* ---- a8  ragged helpers: Models/FCWithEvidences/basic_fc_model.py:80-121 ----
 * offsets[b+1] int32 prefix sum of evidence counts (device). */
/* has[b] (NULL ok) = 1.0 for claims with at least one evidence: row 0 of pad_right(x) is x's first row of the claim times has. */
int gh_seg_offsets(const int64_t* counts, int b, int32_t* offsets, int32_t* pair2claim, int b1, float* has,
                   gh_stream_t stream);
int gh_seg_broadcast(const float* src, const int32_t* pair2claim, float* dst, int b1, int x, gh_stream_t stream);
int gh_seg_sum(const float* src, const int32_t* offsets, float* dst, int b, int x, gh_stream_t stream);
int gh_seg_pad(const float* src, const int32_t* offsets, float* dst, int b, int n_max, int x, int dst_ld,
               gh_stream_t stream);
int gh_seg_unpad(const float* src, const int32_t* offsets, float* dst, int b, int n_max, int x, int src_ld,
                 gh_stream_t stream);
/* ---- evidence-level assembly: graph_based_semantic_structure.py:157-170,195-215 in one launch ----
 * right[b][slot][0:xa] = avg row of the claim's slot-th evidence (zeros beyond its count); right[b][slot][xa:xa+ds] =
 * table[max(sources[b][slot], 0)] (article-source embedding, -1 padding -> row 0; ds = 0: no table);
 * mask[b][slot] = (sum_r document[b][slot][r] >= 1).  sources/document are int32 or int64 (flag).
 * table_rows (ABI 9): rows of `table`; ids are clamped into [0, table_rows) -- nn.Embedding raises for an id beyond the table
 * (graph_based_semantic_structure.py:169), a device kernel cannot, so the event is COUNTED instead (gh_clamp_events). */
int gh_evd_assemble_fwd(const float* avg, const int32_t* offsets, const float* table, int table_rows, const void* sources, int sources_i64,
                        const void* document, int document_i64, int b, int n_max, int xa, int ds, int r,
                        float* right, float* mask, gh_stream_t stream);
/* Backward: d_avg[b1][xa] (NULL ok) = unpad(g[:, :, :xa]); d_table[s][ds] += the g[:, :, xa:] rows of the slots with source s,
 * summed in slot order (deterministic).  g [b][n_max][xa+ds] -- ds is g's row pitch as well: pass the real width even
 * when d_table is NULL (a frozen table just skips the table part).  At most 38 000 slots (b * n_max) per call.
 * EVERY row of d_avg is written (rows of a claim's evidences beyond its n_max-th receive zeros): the caller need not clear it. */
int gh_evd_assemble_bwd(const float* g, const int32_t* offsets, const void* sources, int sources_i64, int table_rows, int b, int n_max,
                        int xa, int ds, float* d_avg, float* d_table, gh_stream_t stream);
/* Out-of-range inputs the kernels clamped for memory safety where the reference would have RAISED (nn.Embedding /
 * nn.CrossEntropyLoss index errors): out_host[0] = labels outside [0, c) seen by gh_cross_entropy, [1] = claim-source ids,
 * [2] = article-source ids outside their table (other than the -1 padding id), [3] = reserved; counted on the CURRENT device since
 * the last reset.  Synchronises the device.  A training loop checks it once per epoch (or per step in debug runs): a non-zero
 * count means corrupt input data was trained on as if it were class / row 0 or the last one. */
int gh_clamp_events(int64_t* out4_host, int reset);
/* masked mean over claim nodes (graph_based_semantic_structure.py:153): dst[b][h] = sum_l hid*mask / len */
int gh_masked_mean_fwd(const float* hid, const int32_t* ids, const float* lens, float* dst, int b, int l, int h,
                       gh_stream_t stream);
int gh_masked_mean_bwd(const float* g, const int32_t* ids, const float* lens, float* dhid, int b, int l, int h,
                       gh_stream_t stream);

/* ---- optimiser: torch.optim.Adam(weight_decay) semantics of Fitting/FittingFC/declare_fitter.py:58-61
 *      on one flat fp32 bucket (the same bucket the RCCL gradient all-reduce uses) ---- */
int gh_adam_step(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float grad_scale, gh_stream_t stream);

/* ---- (e) the path's one exchange step: all-reduce(sum) of the flat fp32 gradient bucket over RCCL / xGMI with a
 *      communicator this library owns (SURVEY 8(b) `flat_allreduce`).  The reference has no distributed code -- its
 *      optimiser is single-process (Fitting/FittingFC/declare_fitter.py:58-61) -- so these calls mirror no reference
 *      interface; they exist so that a non-Python host can run data-parallel steps without
 *      torch.distributed:   rank 0: gh_comm_unique_id -> ship the 128 bytes to every rank by any channel ->
 *      every rank: hipSetDevice, gh_comm_init -> per step: gh_get_backward ..., gh_flat_allreduce(bucket),
 *      gh_adam_step(..., grad_scale = 1 / world).  librccl is dlopen'ed on first use (a copy the process already holds,
 *      e.g. torch's, is reused; $GET_AMD_RCCL overrides the search); nothing else in the library depends on it.
 *      Collectives are enqueued on `stream` and are in place; buffers are device pointers. ---- */
int gh_comm_unique_id(void* id128);                                  /* out: 128 bytes (ncclUniqueId) */
int gh_comm_init(const void* id128, int rank, int world, void** comm);   /* binds to the calling thread's current device */
int gh_comm_destroy(void* comm);
int gh_comm_info(void* comm, int* rank, int* world);                 /* either output may be NULL */
const char* gh_comm_library(void);                                   /* the librccl that was loaded; NULL + gh_last_error() if none */
int gh_flat_allreduce(void* comm, float* buf, int64_t count, gh_stream_t stream);
int gh_flat_broadcast(void* comm, float* buf, int64_t count, int root, gh_stream_t stream);

/* ==== a7  the whole model in two calls: Graph_basedSemantiStructure.forward (graph_based_semantic_structure.py:76-125) ====
 * gh_get_forward / gh_get_backward chain every kernel of the path -- claim cell + masked mean (:144-155), evidence cells
 * with the GSL refinement (:107; wrapper.py:165-172), word-level attention (:173-193), evidence-level assembly and
 * attention (:157-171, :195-221), head (:251-267, :69-74) -- on caller-provided buffers, so that a training step costs
 * the host two library calls instead of ~120 (the per-module entry points above remain the building blocks and the
 * API for callers that use the modules one by one).  Everything is fp32 unless model->storage asks for the bf16 storage
 * pipeline inside the evidence cells; widths: d % 4 == 0, h % 4 == 0, 4 <= d <= h <= 1024 (h > 320: the GSL scorer's projection
 * runs inside gh_scorer_gsl instead of the first cell's last epilogue); the word-embedding table is frozen (as
 * master_get.py:143 constructs it); the claim branch runs on `side_stream` (may equal `stream`) underneath the evidence
 * cells.  Pointers are device pointers; the structs themselves live on the host. */
typedef struct gh_cell_params {            /* one GGNN cell, Models/BiDAF/wrapper.py:177-183 */
  const float *w_p, *w_z0, *w_z1, *w_r0, *w_r1, *w_h0, *w_h1;          /* linear.weight as stored: w_p[h][din], others [h][h] */
  const float *b_z0, *b_z1, *b_r0, *b_r1, *b_h0, *b_h1;
  const float *wt_p, *wt_z0, *wt_z1, *wt_r0, *wt_r1, *wt_h0, *wt_h1;   /* transposes (gh_transpose_batch); backward only */
  float *dw_p, *dw_z0, *dw_z1, *dw_r0, *dw_r1, *dw_h0, *dw_h1;          /* gradients, ACCUMULATED (+=); backward only */
  float *db_z0, *db_z1, *db_r0, *db_r1, *db_h0, *db_h1;
} gh_cell_params;
typedef struct gh_cell_bf16 {              /* bf16 twins of one cell's matrices (bf16 storage mode; gh_weights_refresh makes them) */
  const void *w_p, *w_z0, *w_z1, *w_r0, *w_r1, *w_h0, *w_h1;            /* bf16 [h][din] / [h][h] */
  const void *wt_p, *wt_z0, *wt_z1, *wt_r0, *wt_r1, *wt_h0, *wt_h1;     /* bf16 transposes; backward only */
} gh_cell_bf16;
typedef struct gh_att_params {             /* ConcatNotEqualSelfAtt, thirdparty/two_branches_attention.py:112-148 */
  const float *w1, *w2, *w1t;              /* linear1.weight [ha][xl+dr], linear2.weight [heads][ha], linear1.weight^T */
  float *dw1, *dw2;                        /* ACCUMULATED */
} gh_att_params;
typedef struct gh_get_model {
  int d, h, word_heads, evd_heads, n_classes;       /* embedding width, hidden size, heads, output_size */
  int claim_src_dim, article_src_dim;               /* 0 = that source embedding is not used */
  int claim_src_rows, article_src_rows;             /* rows of the two source tables: claim-source ids are clamped into
                                                       [0, rows) (nn.Embedding raises there; a device kernel cannot) */
  const float* embedding;                           /* [vocab][d] word table (frozen) */
  gh_cell_params claim, cell1, cell2;               /* ggnn4claim_1, ggnn_with_gsl.feat_prop1 / feat_prop2 */
  const float* scorer_w;                            /* ggnn_with_gsl.word_scorer1.proj weight [h] */
  const float* scorer_gate;                         /* its six 1x1 linears packed as gh_scorer_gsl's gate[12] */
  gh_att_params att_word, att_evd;                  /* self_att_word, self_att_evd */
  const float *claim_src_table, *article_src_table; /* [.][claim_src_dim], [.][article_src_dim] or NULL */
  float *d_claim_src_table, *d_article_src_table;   /* ACCUMULATED; NULL = no gradient wanted */
  const float *out0_w, *out0_b, *out0_wt;           /* head: out.0 weight [h][e] (+ transpose [e][h]), bias */
  const float *out1_w, *out1_b, *out1_wt;           /* out.1 weight [n_classes][h] (+ transpose), bias */
  float *d_out0_w, *d_out0_b, *d_out1_w, *d_out1_b; /* ACCUMULATED */
  /* ABI 9 -- BASELINE configs[4]: storage = 1 runs the two EVIDENCE cells in the bf16 storage pipeline of
   * gh_ggnn_cell_fwd_bf16 / gh_ggnn_cell_bwd_bf16 (activations and weights bf16 in HBM, fp32 accumulation, fp32 cell outputs)
   * whenever d % 8 == 0, h % 8 == 0 and the batch has >= 8192 real node rows; everything else stays fp32.  0 = fp32 throughout. */
  int storage;
  const void* embedding16;                          /* bf16 copy of the word table (storage 1) */
  gh_cell_bf16 cell1_16, cell2_16;                  /* bf16 twins of cell1 / cell2 (storage 1) */
  const void *att_word_w1_16, *att_word_w1t_16;     /* bf16 twins of self_att_word.linear1.weight and of its transpose (storage 1; NULL ok:
                                                       the word attention's projections then round fp32 fragments in registers instead) */
} gh_get_model;
typedef struct gh_get_batch {
  int b, b1, l, r, n_max;                           /* claims, pairs, claim length, evidence length, evidence slots per claim */
  int m_real;                                       /* node-compact layout: total real evidence nodes (needs goff..maskf); < 0: padded layout */
  int collapsed;                                    /* node-compact evaluation: the first cell runs on m_real + 1 rows (gh_scorer_gsl pads_collapsed) */
  int k_keep;                                       /* GSL: int(rate * r) */
  const int32_t* q_ids;                             /* [b][l] claim node ids (int32) */
  const void* q_lens; int q_lens_kind;              /* [b] unique claim nodes; kind 0 = float32, 1 = int32, 2 = int64 */
  const uint64_t* q_bits; const float* q_dinv; const float* q_vals;      /* claim graphs (packed adjacency, see top) */
  const int32_t* d_ids;                             /* [b1][r] evidence node ids (padded indexing) */
  const uint64_t* d_bits; const float* d_dinv; const float* d_vals;      /* evidence graphs */
  const int32_t *goff, *rowg, *cids; const float* maskf;                 /* gh_ragged_plan outputs, or all NULL (padded layout) */
  const int64_t* counts;                            /* [b] evidences per claim (sum = b1) */
  int counts_fit;                                   /* (ignored since ABI 8: the backward always zero-fills d_avg) */
  const void* doc_sources; int doc_sources_i64;     /* [b][n_max], -1 = padding slot (article source ids) */
  const void* query_sources; int query_sources_i64; /* [b] claim source ids (claim_src_dim > 0) */
  const void* document; int document_i64;           /* [b][n_max][r] padded evidence node ids (slot mask only) */
  float drop_claim, drop_gnn;                       /* input dropout of the claim cell / of the three evidence-side cells (0 = eval) */
  uint32_t seed_claim, seed_cell1, seed_scorer, seed_cell2;
} gh_get_batch;
/* Buffer plan for (model, batch).  fwd_floats / bwd_floats = sizes (in floats) of the two activation arenas;
 * obs_floats = size of the OBSERVABLES buffer -- a few MB holding what a caller keeps after the step (ABI 7: a buffer of
 * its own, so that holding the logits / attention weights / keep-sets does not pin the multi-GB forward arena). */
typedef struct gh_get_plan {
  int64_t fwd_floats, bwd_floats, obs_floats;
  int64_t phi, word_w, evd_w, score, keep;          /* offsets (floats) of the observable results inside the observables buffer:
                                                       phi [b][n_classes]; word_w [rows][word_heads] (rows = m_real, or b1*r padded);
                                                       evd_w [b][n_max][evd_heads]; score [b1][r]; keep [b1][W] uint64 (8-byte aligned) */
} gh_get_plan;
int gh_get_plan_buffers(const gh_get_model* model, const gh_get_batch* batch, gh_get_plan* plan);
/* sizeof of {gh_get_model, gh_get_batch, gh_get_plan, gh_cell_params, gh_cell_bf16}: lets a binding verify its mirror of the structs. */
int gh_get_struct_sizes(int64_t* out5_host);
/* Forward into `arena_fwd` (plan.fwd_floats floats) and `obs` (plan.obs_floats floats); both 256-byte aligned and kept until
 * the backward has run (evaluation: the arena may be released as soon as the call has been issued and the stream drained). */
int gh_get_forward(const gh_get_model* model, const gh_get_batch* batch, float* arena_fwd, float* obs, gh_stream_t stream,
                   gh_stream_t side_stream);
/* Backward from g_phi [b][n_classes] (+ optional g_word_w / g_evd_w, shaped like the forward's weights outputs, or NULL).
 * phase 0: everything; 1: down to and including the second evidence cell -- every gradient outside the first evidence cell
 * and the claim branch is final in stream order when it returns (the data-parallel early all-reduce starts here);
 * 2: the rest (first evidence cell; joins the side stream).  Gradients are ACCUMULATED into model->d*. */
int gh_get_backward(const gh_get_model* model, const gh_get_batch* batch, const float* arena_fwd, const float* obs, float* arena_bwd,
                    const float* g_phi, const float* g_word_w, const float* g_evd_w, int phase,
                    gh_stream_t stream, gh_stream_t side_stream);
/* Mean cross-entropy of logits [b][c] against labels [b] (int64, clamped into [0, c): no ignore_index), fused with its gradient (losses.py:29-32 CrossEntropyLoss):
 * loss[0] = mean_b(logsumexp(phi_b) - phi_b[y_b]); dphi [b][c] = (softmax(phi_b) - onehot(y_b)) / b. */
int gh_cross_entropy(const float* phi, const int64_t* labels, int b, int c, float* loss, float* dphi, gh_stream_t stream);
/* Batch preparation in one call (interactions.py:334-351 for both sides + basic_fc_model.py:94-121's padded document):
 * graph build of the b claims and b1 evidences, node-compact plan (m_real must be known to the host; < 0 = skip the plan)
 * and the scatter of the evidence node ids into document[b][n_max][r] (slot[b1] = row of each pair; rows of unused slots
 * must be zero already and stay untouched). */
int gh_get_prepare(const int32_t* claim_tokens, const int32_t* claim_len, int b, int l,
                   const int32_t* evd_tokens, const int32_t* evd_len, int b1, int r, int window,
                   int32_t* q_ids, int32_t* q_n, uint64_t* q_bits, float* q_dinv,
                   int32_t* d_ids, int32_t* d_n, uint64_t* d_bits, float* d_dinv,
                   int m_real, int32_t* goff, int32_t* rowg, int32_t* src, int32_t* cids, float* maskf,
                   const int64_t* slot, int32_t* document, gh_stream_t stream);

/* Everything the forward / backward need that is DERIVED from weight matrices, for n matrices in one launch (after the
 * optimiser step): dst_t[i] = src[i]^T as fp32 [cols][rows] (the backward's dX operands), and -- bf16 storage mode, any of the
 * two arrays or any entry may be NULL -- dst_w16[i] = bf16(src[i]) [rows][cols], dst_t16[i] = bf16(src[i]^T) [cols][rows]. */
int gh_weights_refresh(int n, const void* const* src, void* const* dst_t, void* const* dst_w16, void* const* dst_t16,
                       const int* rows, const int* cols, gh_stream_t stream);

/* ---- measurement hook (bench.py): HIP events around every kernel launch on its own stream ----
 * rows of `out` (each {total ms, total algorithmic work, launches}; work = flops for GEMMs, bytes otherwise):
 * 0 activation-sized GEMM launches (>= 8192 rows) NT/NN, 1 same TN, 2 few-row GEMM launches NT/NN, 3 same TN, 4 spmm, 5 scorer_gsl,
 * 6 graph_build, 7 att_softmax_fwd, 8 att_softmax_bwd, 9 att_dpre, 10 gate_bwd_pre, 11 colsum, 12 adam */
#define GH_PROFILE_ROWS 13
int gh_profile_enable(int on);
/* Restrict the instrumentation to the rows whose bit is set in tag_mask (default: all).  Two events per launch cost
 * ~1.5 us of queue bubbles each, ~0.4 ms per bench step when every kernel is instrumented: bench.py times the step
 * with only the dominant kernel's row selected and collects the full table in an extra, untimed pass. */
int gh_profile_select(uint32_t tag_mask);
int gh_profile_collect(double* out_host, int rows);

/* GEMM launches since the last reset: out_host[0] on the float4 fast kernel, [1] on the generic (scalar-guarded) kernel,
 * [2] on the generic kernel with >= 1 GFLOP of work -- the last one should stay 0 in a healthy training step (a large
 * GEMM only falls back when an operand is misaligned or oddly shaped). */
int gh_gemm_path_counters(int64_t* out_host, int reset);

#ifdef __cplusplus
}
#endif
#endif /* GET_HIP_H */
