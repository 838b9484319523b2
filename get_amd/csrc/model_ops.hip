// Composite entry points: the whole GET forward / backward as ONE library call each (include/get_hip.h "a7"), the fused
// cross-entropy and the one-call batch preparation.  Host-side chaining of the fused building blocks of gemm_ops.hip /
// graph_ops.hip / misc_ops.hip on two streams, plus the few small kernels that replace the at::native glue the
// module-by-module path needed (lens casts, `* has`, torch.cat, index_copy_, the CE kernels).
//
// Why: issuing a training step module by module costs ~120 Python -> ctypes -> autograd round trips (~2.2 ms of host
// time); at realistic evidence counts (~220 pairs per step, ~1.3 ms of device time) the step was bound by that.
#include "../../include/get_hip.h"
#include "common.h"
#include <mutex>
#include <unordered_map>

namespace gh {

// ---------------------------------------------------------------------------------------------- buffer plan
struct Dims {
  int B, B1, L, R, n, D, H, hw, he, C, cs, as, Xl, Xa, Dre, E, W;
  bool compact;
  int Mr, Mt, M1, Mq;
  bool bf;            // the two evidence cells run the bf16 storage pipeline (model->storage = 1 and the shapes qualify)
  bool att16;         // bf16 storage and the word attention runs on the bf16 twins (gemm mode 1, twins of linear1.weight at hand): the second
                      // cell's output exists as bf16 rows only -- no fp32 copy (190 MB at h = 768), the attention's streaming kernels read the bf16 rows
  bool fuse_scorer;   // the GSL scorer's projection rides in the first cell's last epilogue (whole rows in <= 2 column blocks: h <= 320)
};
// out32: the fp32 cell output (== out in fp32 storage; a buffer of its own beside the bf16 twin `out` in bf16 storage)
struct CellBuf { int64_t xp, a, z, rr, rx, hh, out, out32, xdrop; };      // xdrop: bf16 storage + dropout: the projection's masked operand rows (-1: none)
struct FwdBuf {
  int64_t offsets, pair2claim, has, lens_eff, rowc, maskf_p;
  CellBuf q, c1, c2;
  int64_t q_repr, score_x, score, keep;
  int64_t uw, tw, ew, ww, avg, new_left;
  int64_t right_e, mask_e, ue, te, ee, we, att_e, y0, phi;
  int64_t total;
  int64_t obs_total;      // phi, ww, we, score, keep are offsets into the SEPARATE observables buffer (gh_get_plan::obs_floats)
};
struct BwdBuf {
  int64_t d_y0, d_new_left, d_att_e, de_e, dpre_e, du_e, dright_e, d_avg;
  int64_t de_w, dw_w, dpre_w, du_w, du_c, d_q, g2, d_qhid, qs[5], sc2[5], sc1[5], dx2, dw2p_e, dw2p_w, dw_e;
  int64_t total;
};
struct Bump {
  int64_t off = 0;
  int64_t take(int64_t nfloats) { const int64_t o = off; off += (nfloats + 63) & ~(int64_t)63; return o; }     // 256-byte slots
};

static int make_dims(const gh_get_model* Mo, const gh_get_batch* Ba, Dims& d) {
  GH_REQUIRE(Mo && Ba, "get: NULL model / batch descriptor");
  d.B = Ba->b; d.B1 = Ba->b1; d.L = Ba->l; d.R = Ba->r; d.n = Ba->n_max;
  d.D = Mo->d; d.H = Mo->h; d.hw = Mo->word_heads; d.he = Mo->evd_heads; d.C = Mo->n_classes;
  d.cs = Mo->claim_src_dim; d.as = Mo->article_src_dim;
  GH_REQUIRE(d.B > 0 && d.B1 > 0 && d.L > 0 && d.R > 0 && d.n > 0, "get: bad batch sizes b=%d b1=%d l=%d r=%d n_max=%d", d.B, d.B1, d.L, d.R, d.n);
  GH_REQUIRE(d.R <= 256 && d.L <= 256, "get: padded graph sizes above 256 nodes are not supported (l=%d r=%d)", d.L, d.R);
  GH_REQUIRE(d.D % 4 == 0 && d.H % 4 == 0 && d.D >= 4 && d.D <= d.H && d.H <= 1024,
             "get: the composite path needs float4-shaped widths with d <= h <= 1024 (d=%d h=%d); use the per-module entry points", d.D, d.H);
  GH_REQUIRE(Mo->storage == 0 || Mo->storage == 1, "get: model->storage %d not in {0 (fp32), 1 (bf16 storage in the evidence cells)}", Mo->storage);
  GH_REQUIRE(d.hw >= 1 && d.hw <= 8 && d.he >= 1 && d.he <= 8, "get: heads %d / %d not in [1,8]", d.hw, d.he);
  GH_REQUIRE(Ba->k_keep >= 0 && Ba->k_keep <= d.R, "get: k_keep=%d not in [0, r=%d]", Ba->k_keep, d.R);
  GH_REQUIRE(d.C >= 1 && d.cs >= 0 && d.as >= 0 && d.cs % 4 == 0 && d.as % 4 == 0, "get: bad class / source widths (%d, %d, %d)", d.C, d.cs, d.as);
  d.Xl = d.H + d.cs; d.Xa = d.H * d.hw; d.Dre = d.Xa + d.as; d.E = d.Xl + d.Dre * d.he; d.W = words_for(d.R);
  d.compact = Ba->m_real >= 0;
  d.Mt = d.B1 * d.R;
  if (d.compact) {
    GH_REQUIRE(Ba->goff && Ba->rowg && Ba->cids && Ba->maskf, "get: the node-compact layout needs goff, rowg, cids and maskf");
    GH_REQUIRE(Ba->m_real > 0 && Ba->m_real <= d.Mt, "get: m_real=%d does not fit b1*r=%d", Ba->m_real, d.Mt);
    d.Mr = Ba->m_real;
    d.M1 = Ba->collapsed ? (d.Mr + 1 < d.Mt ? d.Mr + 1 : d.Mt) : d.Mt;
    GH_REQUIRE(!(Ba->collapsed && Ba->drop_gnn > 0.f), "get: collapsed padding rows are an evaluation-mode layout (no dropout)");
  } else {
    d.Mr = d.Mt; d.M1 = d.Mt;
  }
  d.Mq = d.B * d.L;
  d.fuse_scorer = d.H <= 320;
  // same rule as the per-module path (get_amd/ops.py bf16_cell_ok): 16-byte bf16 rows and activation-sized launches
  d.bf = Mo->storage == 1 && d.D % 8 == 0 && d.H % 8 == 0 && d.Mr >= 8192;
  if (d.bf) {
    const gh_cell_bf16* cs[2] = {&Mo->cell1_16, &Mo->cell2_16};
    for (const gh_cell_bf16* c : cs)
      GH_REQUIRE(c->w_p && c->w_z0 && c->w_z1 && c->w_r0 && c->w_r1 && c->w_h0 && c->w_h1,
                 "get: storage = 1 needs the bf16 twins of both evidence cells' weights (gh_weights_refresh)");
    GH_REQUIRE(Mo->embedding16, "get: storage = 1 needs the bf16 copy of the word table (embedding16)");
  }
  static int att16_on = -1;
  if (att16_on < 0) att16_on = measure_env("GH_ATT16", 1);
  // att16 = "no fp32 copy of the second cell's output exists": it must imply every condition of att_fwd_impl's / att_bwd_impl's own
  // use16 predicate (gemm_ops.hip: bf16 gemm mode, >= 8192 rows, right / left / hidden widths multiples of 8) -- restated here in
  // full rather than relied upon by construction (d.bf already holds Mr >= 8192 and H % 8 == 0; the word attention's right and
  // left widths and its hidden width are all H)
  d.att16 = d.bf && gemm_mode() == 1 && Mo->att_word_w1_16 && Mo->att_word_w1t_16 && att16_on && d.Mr >= 8192 && d.H % 8 == 0;
  return 0;
}

static void cell_buf(Bump& b, CellBuf& c, int64_t rows, int H, bool bf = false, bool need32 = true, int64_t drop_din = 0) {
  c.xdrop = (bf && drop_din > 0) ? b.take(rows * drop_din / 2) : -1;
  const int64_t e = bf ? rows * H / 2 : rows * H;      // floats per saved tensor (bf16 storage: two values per float slot)
  c.xp = b.take(e); c.a = b.take(e); c.z = b.take(e); c.rr = b.take(e);
  c.rx = b.take(e); c.hh = b.take(e); c.out = b.take(e);
  c.out32 = bf ? (need32 ? b.take(rows * H) : -1) : c.out;      // (-1: no fp32 consumer -- the first cell when its scorer projection is fused)
}

static int layout(const gh_get_model* Mo, const gh_get_batch* Ba, Dims& d, FwdBuf& f, BwdBuf& w) {
  if (int e = make_dims(Mo, Ba, d)) return e;
  Bump b;
  f.offsets = b.take(d.B + 1); f.pair2claim = b.take(d.B1); f.has = b.take(d.B); f.lens_eff = b.take(d.B);
  f.rowc = b.take(d.Mr);
  f.maskf_p = d.compact ? -1 : b.take(d.Mt);
  // observables (small; what the caller keeps after the step) live in their own buffer, so that holding on to the logits,
  // attention weights, scores or keep-sets does not pin the multi-GB activation arena
  {
    Bump o;
    f.phi = o.take((int64_t)d.B * d.C);
    f.ww = o.take((int64_t)d.Mr * d.hw);
    f.we = o.take((int64_t)d.B * d.n * d.he);
    f.score = o.take((int64_t)d.B1 * d.R);
    f.keep = o.take((int64_t)d.B1 * d.W * 2);
    f.obs_total = o.off;
  }
  cell_buf(b, f.q, d.Mq, d.H);
  f.q_repr = b.take((int64_t)d.B * d.H);
  // bf16 storage: the first cell's fp32 output would feed the scorer's projection only, which the cell's last epilogue computes
  // itself (per column block when h > 320: up to 8 partials) -- no fp32 copy of that output exists then
  const bool pre_drop = d.bf && Ba->drop_gnn > 0.f;
  cell_buf(b, f.c1, d.M1, d.H, d.bf, !d.bf, pre_drop ? d.D : 0);
  f.score_x = b.take((int64_t)d.M1 * ((d.fuse_scorer || !d.bf) ? 1 : 8));
  cell_buf(b, f.c2, d.Mr, d.H, d.bf, !d.att16, pre_drop ? d.H : 0);
  f.uw = b.take((int64_t)d.B * d.H); f.tw = b.take((int64_t)d.Mr * d.H); f.ew = b.take((int64_t)d.Mr * d.hw);
  f.avg = b.take((int64_t)d.B1 * d.Xa);
  f.new_left = d.cs > 0 ? b.take((int64_t)d.B * d.Xl) : f.q_repr;
  f.right_e = b.take((int64_t)d.B * d.n * d.Dre); f.mask_e = b.take((int64_t)d.B * d.n);
  f.ue = b.take((int64_t)d.B * d.H); f.te = b.take((int64_t)d.B * d.n * d.H); f.ee = b.take((int64_t)d.B * d.n * d.he);
  f.att_e = b.take((int64_t)d.B * d.Dre * d.he);
  f.y0 = b.take((int64_t)d.B * d.H);
  f.total = b.off;
  Bump c;
  w.d_y0 = c.take((int64_t)d.B * d.H); w.d_new_left = c.take((int64_t)d.B * d.Xl); w.d_att_e = c.take((int64_t)d.B * d.Dre * d.he);
  w.de_e = c.take((int64_t)d.B * d.n * d.he); w.dpre_e = c.take((int64_t)d.B * d.n * d.H); w.du_e = c.take((int64_t)d.B * d.H);
  w.dright_e = c.take((int64_t)d.B * d.n * d.Dre);
  w.d_avg = c.take((int64_t)d.B1 * d.Xa);
  w.de_w = c.take((int64_t)d.Mr * d.hw); w.dw_w = c.take((int64_t)((d.H + 511) / 512) * d.Mr * d.hw);      // (one partial dw per 512-float column range of a row)
  w.dw_e = c.take((int64_t)((d.Dre + 511) / 512) * d.B * d.n * d.he); w.dpre_w = c.take((int64_t)d.Mr * d.H); w.du_w = c.take((int64_t)d.B1 * d.H);
  w.du_c = c.take((int64_t)d.B * d.H);
  w.d_q = d.cs > 0 ? c.take((int64_t)d.B * d.H) : w.d_new_left;
  w.g2 = c.take((int64_t)d.Mr * d.H);
  w.d_qhid = c.take((int64_t)d.Mq * d.H);
  for (int i = 0; i < 5; ++i) w.qs[i] = c.take((int64_t)d.Mq * d.H);
  // one scratch set per evidence cell: the weight-gradient stream still reads the second cell's dzp / drp / dhp / dxp
  // while the main stream already runs the first cell's chain
  // (bf16 storage: the scratch holds bf16; dx2 -- the fp32 gradient between the two cells -- exists only there, the fp32
  //  pipeline writes it straight into the first cell's gate head)
  const int64_t se = d.bf ? (int64_t)d.Mr * d.H / 2 : (int64_t)d.Mr * d.H;
  for (int i = 0; i < 5; ++i) w.sc2[i] = c.take(se);
  for (int i = 0; i < 5; ++i) w.sc1[i] = c.take(se);
  w.dx2 = d.bf ? c.take((int64_t)d.Mr * d.H) : w.g2;
  // per-pair partials of the two attention layers' dW2 (reduced on the side stream)
  w.dw2p_e = c.take((int64_t)d.B * d.he * d.H); w.dw2p_w = c.take((int64_t)d.B1 * d.hw * d.H);
  w.total = c.off;
  return 0;
}

// ---------------------------------------------------------------------------------------------- stream fork / join events
struct DevEvents { hipEvent_t ev[12]; };
static std::mutex g_ev_mu;
static std::unordered_map<int, DevEvents> g_events;
static int get_events(DevEvents& out) {
  int dev = 0;
  GH_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_ev_mu);
  auto it = g_events.find(dev);
  if (it == g_events.end()) {
    DevEvents e;
    for (int i = 0; i < 12; ++i) GH_CHECK_HIP(hipEventCreateWithFlags(&e.ev[i], hipEventDisableTiming));
    it = g_events.emplace(dev, e).first;
  }
  out = it->second;
  return 0;
}
// `to` waits for everything issued on `from` so far
static int stream_after(hipStream_t to, hipStream_t from, hipEvent_t ev) {
  if (to == from) return 0;
  GH_CHECK_HIP(hipEventRecord(ev, from));
  GH_CHECK_HIP(hipStreamWaitEvent(to, ev, 0));
  return 0;
}

// ---------------------------------------------------------------------------------------------- small glue kernels
// rowc[row] = claim of feature row `row` (row -> pair through rowg, or row / R in the padded layout; pair -> claim);
// maskf_p[row] = (id >= 1) for the padded layout; lens_eff[b] = len[b] / has[b] (+inf for a claim without evidences), so
// that gh_masked_mean_*'s 1 / lens scale also carries the `* has` of graph_based_semantic_structure.py:213 / pad_right.
__global__ void __launch_bounds__(256)
rows_prep_kernel(const int32_t* __restrict__ pair2claim, const int32_t* __restrict__ rowg, int R, int Mr,
                 int32_t* __restrict__ rowc, const int32_t* __restrict__ d_ids, float* __restrict__ maskf_p,
                 const void* __restrict__ q_lens, int kind, const float* __restrict__ has, float* __restrict__ lens_eff, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Mr) {
    const int pair = rowg ? rowg[i] : i / R;
    rowc[i] = pair2claim[pair];
    if (maskf_p) maskf_p[i] = d_ids[i] >= 1 ? 1.f : 0.f;
  }
  if (i < B) {
    float len;
    if (kind == 0) len = reinterpret_cast<const float*>(q_lens)[i];
    else if (kind == 1) len = (float)reinterpret_cast<const int32_t*>(q_lens)[i];
    else len = (float)reinterpret_cast<const int64_t*>(q_lens)[i];
    lens_eff[i] = has[i] > 0.f ? len : INFINITY;
  }
}

// seg_offsets_kernel + rows_prep_kernel in one launch for B <= 1024 claims: every workgroup scans the counts itself (LDS), workgroup
// 0 writes offsets / pair2claim / has / lens_eff as the two kernels did, every thread finds its row's claim by a binary search.
template <typename TL>
__global__ void __launch_bounds__(256)
seg_rows_kernel(const int64_t* __restrict__ counts, int B, int b1, const int32_t* __restrict__ rowg, int R, int Mr,
                int32_t* __restrict__ offsets, int32_t* __restrict__ pair2claim, float* __restrict__ has, int32_t* __restrict__ rowc,
                const int32_t* __restrict__ d_ids, float* __restrict__ maskf_p, const TL* __restrict__ q_lens, float* __restrict__ lens_eff) {
  __shared__ int off[1025];
  const int tid = threadIdx.x;
  // inclusive scan of the (few) counts: thread t owns claims [4 t, 4 t + 4)
  int c[4], sum = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int b = 4 * tid + k; c[k] = b < B ? (int)counts[b] : 0; sum += c[k]; }
  __shared__ int part[256];
  part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - sum;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int b = 4 * tid + k; if (b < B) off[b] = run; run += c[k]; }
  if (tid == 255) off[B] = part[255];
  __syncthreads();
  const int total = off[B];
  if (blockIdx.x == 0) {
    for (int b = tid; b <= B; b += 256) offsets[b] = off[b];
    for (int b = tid; b < B; b += 256) {
      const float hb = counts[b] > 0 ? 1.f : 0.f;
      has[b] = hb;
      lens_eff[b] = hb > 0.f ? (float)q_lens[b] : INFINITY;
    }
  }
  auto claim_of = [&](int p) {      // largest b with off[b] <= p  (pairs beyond the counts' total map to claim 0, as seg_offsets does)
    if (p >= total) return 0;
    int lo = 0, hi = B - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (off[mid] <= p) lo = mid; else hi = mid - 1; }
    return lo;
  };
  const int i = blockIdx.x * 256 + tid;
  if (i < b1) pair2claim[i] = claim_of(i);
  if (i < Mr) {
    const int pair = rowg ? rowg[i] : i / R;
    rowc[i] = claim_of(pair);
    if (maskf_p) maskf_p[i] = d_ids[i] >= 1 ? 1.f : 0.f;
  }
}

// new_left[b] = [claim_src_table[src[b]] * has[b] | q_repr[b]]   (q_repr already carries has; graph_based_semantic_structure.py:113-116)
template <typename TS>
__global__ void __launch_bounds__(256)
left_assemble_fwd_kernel(const float* __restrict__ table, const TS* __restrict__ src, const float* __restrict__ q_repr,
                         const float* __restrict__ has, float* __restrict__ new_left, int cs, int H, int rows,
                         unsigned int* __restrict__ clamped) {
  const int b = blockIdx.x;
  const float hb = has[b];
  const long long sb = (long long)src[b];
  if ((sb < 0 || sb >= rows) && threadIdx.x == 0) atomicAdd(clamped + 1, 1u);      // nn.Embedding would raise (:113); counted, gh_clamp_events
  const float* tp = table + (size_t)(sb < 0 ? 0 : (sb >= rows ? rows - 1 : sb)) * cs;
  float* o = new_left + (size_t)b * (cs + H);
  for (int i = threadIdx.x; i < cs; i += blockDim.x) o[i] = tp[i] * hb;
  for (int i = threadIdx.x; i < H; i += blockDim.x) o[cs + i] = q_repr[(size_t)b * H + i];
}
// d_q[b] = d_new_left[b][cs:] ; d_table[src[b]] += d_new_left[b][:cs] * has[b], claims walked in order by ONE workgroup
// (duplicate sources inside a batch add deterministically, no atomics)
template <typename TS>
__global__ void __launch_bounds__(256)
left_assemble_bwd_kernel(const float* __restrict__ d_new_left, const TS* __restrict__ src, const float* __restrict__ has,
                         float* __restrict__ d_q, float* __restrict__ d_table, int B, int cs, int H, int rows) {
  for (int i = threadIdx.x; i < B * H; i += blockDim.x) {
    const int b = i / H, c = i - b * H;
    d_q[i] = d_new_left[(size_t)b * (cs + H) + cs + c];
  }
  if (!d_table) return;
  for (int c = threadIdx.x; c < cs; c += blockDim.x)
    for (int b = 0; b < B; ++b) {
      const long long sb = (long long)src[b];
      d_table[(size_t)(sb < 0 ? 0 : (sb >= rows ? rows - 1 : sb)) * cs + c] += d_new_left[(size_t)b * (cs + H) + c] * has[b];
    }
}

// losses.py:29-32: mean CE and its gradient in one pass.  One workgroup; claims strided over the threads; the loss is
// reduced in a fixed order.
__global__ void __launch_bounds__(256)
cross_entropy_kernel(const float* __restrict__ phi, const int64_t* __restrict__ labels, int B, int C, float* __restrict__ loss,
                     float* __restrict__ dphi, unsigned int* __restrict__ clamped) {
  __shared__ float part[256];
  float acc = 0.f;
  const float invb = 1.f / (float)B;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float* p = phi + (size_t)b * C;
    float mx = p[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
    const float lse = mx + logf(se);
    const long long yl = labels[b];
    const int y = yl < 0 ? 0 : (yl >= C ? C - 1 : (int)yl);      // (no ignore_index; out-of-range labels are clamped ...
    if (yl < 0 || yl >= C) atomicAdd(clamped, 1u);               //  ... and counted: nn.CrossEntropyLoss raises there, gh_clamp_events)
    acc += lse - p[y];
    for (int c = 0; c < C; ++c) dphi[(size_t)b * C + c] = (expf(p[c] - lse) - (c == y ? 1.f : 0.f)) * invb;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = part[0] * invb;
}

// document[slot[g]][:] = d_ids[g][:]   (padded layout; the node-compact plan's fill kernel does it on the way)
__global__ void __launch_bounds__(256)
document_scatter_kernel(const int32_t* __restrict__ d_ids, const int64_t* __restrict__ slot, int32_t* __restrict__ document, int R) {
  const int g = blockIdx.x;
  const size_t dst = (size_t)slot[g] * R;
  for (int j = threadIdx.x; j < R; j += blockDim.x) document[dst + j] = d_ids[(size_t)g * R + j];
}

}  // namespace gh

using namespace gh;

extern "C" int gh_get_plan_buffers(const gh_get_model* Mo, const gh_get_batch* Ba, gh_get_plan* P) {
  GH_REQUIRE(P, "get_plan_buffers: NULL plan");
  Dims d; FwdBuf f; BwdBuf w;
  if (int e = layout(Mo, Ba, d, f, w)) return e;
  P->fwd_floats = f.total; P->bwd_floats = w.total; P->obs_floats = f.obs_total;
  P->phi = f.phi; P->word_w = f.ww; P->evd_w = f.we; P->score = f.score; P->keep = f.keep;
  return 0;
}

extern "C" int gh_get_struct_sizes(int64_t* out) {
  GH_REQUIRE(out, "get_struct_sizes: NULL");
  out[0] = sizeof(gh_get_model); out[1] = sizeof(gh_get_batch); out[2] = sizeof(gh_get_plan); out[3] = sizeof(gh_cell_params);
  out[4] = sizeof(gh_cell_bf16);
  return 0;
}

#define GH_TRY(expr) do { if (int _rc = (expr)) return _rc; } while (0)

static inline const int32_t* I32(const float* A, int64_t off) { return reinterpret_cast<const int32_t*>(A + off); }
static inline int32_t* I32(float* A, int64_t off) { return reinterpret_cast<int32_t*>(A + off); }

// c16 != NULL: bf16 storage pipeline -- x (or the table behind ids) and the saved tensors hold bf16, the weights come from the twins
static int cell_fwd(const gh_cell_params& c, const gh_cell_bf16* c16, const CellBuf& cb, float* A, const uint64_t* bits, const float* dinv,
                    const float* vals, const uint64_t* keep, const int32_t* goff, int m_real, int m_rows, const float* x, const int32_t* ids,
                    int n, int r, int din, int h, float drop_p, uint32_t seed, const float* score_w, float* score_x, float sdrop,
                    uint32_t sseed, hipStream_t s, int pad_out_dead = 0, int* score_parts = nullptr) {
  typedef const float* cf;
  if (c16)
    return cell_fwd_impl(1, cb.out32 >= 0 ? A + cb.out32 : nullptr, bits, dinv, vals, keep, goff, m_real, m_rows, x, ids, n, r, din, h, (cf)c16->w_p, (cf)c16->w_z0,
                         (cf)c16->w_z1, (cf)c16->w_r0, (cf)c16->w_r1, (cf)c16->w_h0, (cf)c16->w_h1, c.b_z0, c.b_z1, c.b_r0, c.b_r1, c.b_h0,
                         c.b_h1, A + cb.xp, A + cb.a, A + cb.z, A + cb.rr, A + cb.rx, A + cb.hh, A + cb.out, drop_p, seed, score_w, score_x,
                         sdrop, sseed, (void*)s, 0, score_parts, cb.xdrop >= 0 ? A + cb.xdrop : nullptr);
  return cell_fwd_impl(0, nullptr, bits, dinv, vals, keep, goff, m_real, m_rows, x, ids, n, r, din, h, c.w_p, c.w_z0, c.w_z1, c.w_r0,
                       c.w_r1, c.w_h0, c.w_h1, c.b_z0, c.b_z1, c.b_r0, c.b_r1, c.b_h0, c.b_h1, A + cb.xp, A + cb.a, A + cb.z,
                       A + cb.rr, A + cb.rx, A + cb.hh, A + cb.out, drop_p, seed, score_w, score_x, sdrop, sseed, (void*)s, pad_out_dead);
}
static int cell_bwd(const gh_cell_params& c, const gh_cell_bf16* c16, const CellBuf& cb, const float* A, const uint64_t* bits, const float* dinv,
                    const float* vals, const uint64_t* keep, const int32_t* goff, int m_real, const float* x, const int32_t* ids, int n,
                    int r, int din, int h, const float* g, float* W, const int64_t* sc, float* dx, float drop_p, uint32_t seed,
                    hipStream_t s, int pre_done = 0, const GateFuse* next = nullptr) {
  GH_REQUIRE(c.dw_p && c.db_z0 && c.db_z1, "get_backward: a cell's gradient outputs are missing");
  if (c16) {
    typedef const float* cf;
    GH_REQUIRE(c16->wt_p && c16->wt_z0 && c16->wt_z1 && c16->wt_r0 && c16->wt_r1 && c16->wt_h0 && c16->wt_h1,
               "get_backward: storage = 1 needs the bf16 twins of the cells' transposed weights (gh_weights_refresh)");
    return cell_bwd_impl(1, bits, dinv, vals, keep, goff, m_real, x, ids, n, r, din, h, (cf)c16->wt_p, (cf)c16->wt_z0, (cf)c16->wt_z1,
                         (cf)c16->wt_r0, (cf)c16->wt_r1, (cf)c16->wt_h0, (cf)c16->wt_h1, A + cb.xp, A + cb.a, A + cb.z, A + cb.rr,
                         A + cb.rx, A + cb.hh, g, W + sc[0], W + sc[1], W + sc[2], W + sc[3], W + sc[4], dx, c.dw_p, c.dw_z0, c.dw_z1,
                         c.dw_r0, c.dw_r1, c.dw_h0, c.dw_h1, c.db_z0, c.db_r0, c.db_h0, c.db_z1, c.db_r1, c.db_h1, drop_p, seed, (void*)s,
                         nullptr, nullptr, nullptr, pre_done, next, cb.xdrop >= 0 ? A + cb.xdrop : nullptr);
  }
  GH_REQUIRE(c.wt_p, "get_backward: a cell's transposes are missing");
  return cell_bwd_impl(0, bits, dinv, vals, keep, goff, m_real, x, ids, n, r, din, h, c.wt_p, c.wt_z0, c.wt_z1, c.wt_r0, c.wt_r1,
                       c.wt_h0, c.wt_h1, A + cb.xp, A + cb.a, A + cb.z, A + cb.rr, A + cb.rx, A + cb.hh, g, W + sc[0], W + sc[1],
                       W + sc[2], W + sc[3], W + sc[4], dx, c.dw_p, c.dw_z0, c.dw_z1, c.dw_r0, c.dw_r1, c.dw_h0, c.dw_h1, c.db_z0,
                       c.db_r0, c.db_h0, c.db_z1, c.db_r1, c.db_h1, drop_p, seed, (void*)s, nullptr, nullptr, nullptr, pre_done, next);
}

extern "C" int gh_get_forward(const gh_get_model* Mo, const gh_get_batch* Ba, float* A, float* O, gh_stream_t stream, gh_stream_t side_stream) {
  Dims d; FwdBuf f; BwdBuf w;
  GH_TRY(layout(Mo, Ba, d, f, w));
  GH_REQUIRE(A && (reinterpret_cast<uintptr_t>(A) & 255) == 0, "get_forward: the arena must be 256-byte aligned");
  GH_REQUIRE(O && (reinterpret_cast<uintptr_t>(O) & 255) == 0, "get_forward: the observables buffer must be 256-byte aligned");
  GH_REQUIRE(Ba->q_ids && Ba->q_lens && Ba->q_bits && (Ba->q_dinv || Ba->q_vals) && Ba->d_ids && Ba->d_bits && (Ba->d_dinv || Ba->d_vals) &&
             Ba->counts && Ba->document, "get_forward: a batch tensor is missing (ids, lens, packed graphs, counts, document)");
  GH_REQUIRE(Ba->q_lens_kind >= 0 && Ba->q_lens_kind <= 2, "get_forward: q_lens_kind %d not in {0,1,2}", Ba->q_lens_kind);
  GH_REQUIRE(d.as == 0 || Ba->doc_sources, "get_forward: article-source ids are missing");
  GH_REQUIRE(d.cs == 0 || Ba->query_sources, "get_forward: claim-source ids are missing");
  GH_REQUIRE(Mo->embedding && Mo->scorer_w && Mo->scorer_gate && Mo->out0_w && Mo->out1_w, "get_forward: missing model tensors");
  GH_REQUIRE((d.cs == 0) == (Mo->claim_src_table == nullptr) && (d.as == 0) == (Mo->article_src_table == nullptr),
             "get_forward: source tables and their widths must come together");
  GH_REQUIRE(d.cs == 0 || Mo->claim_src_rows > 0, "get_forward: claim_src_rows must give the claim-source table's row count");
  GH_REQUIRE(d.as == 0 || Mo->article_src_rows > 0, "get_forward: article_src_rows must give the article-source table's row count");
  GH_REQUIRE(Ba->drop_claim >= 0.f && Ba->drop_claim < 1.f && Ba->drop_gnn >= 0.f && Ba->drop_gnn < 1.f, "get_forward: dropout p not in [0,1)");
  hipStream_t s = (hipStream_t)stream, ss = side_stream ? (hipStream_t)side_stream : s;
  DevEvents ev;
  GH_TRY(get_events(ev));
  const int H = d.H;
  // ---- claim -> pair map, row -> claim map, masks
  if (d.B <= 1024) {
    const int nthr = d.Mr > d.B1 ? d.Mr : d.B1;
    const dim3 grid((nthr + 255) / 256);
    const int32_t* rg = d.compact ? Ba->rowg : nullptr;
    float* mp = d.compact ? nullptr : A + f.maskf_p;
    if (Ba->q_lens_kind == 0)
      hipLaunchKernelGGL(seg_rows_kernel<float>, grid, dim3(256), 0, s, Ba->counts, d.B, d.B1, rg, d.R, d.Mr, I32(A, f.offsets), I32(A, f.pair2claim),
                         A + f.has, I32(A, f.rowc), Ba->d_ids, mp, (const float*)Ba->q_lens, A + f.lens_eff);
    else if (Ba->q_lens_kind == 1)
      hipLaunchKernelGGL(seg_rows_kernel<int32_t>, grid, dim3(256), 0, s, Ba->counts, d.B, d.B1, rg, d.R, d.Mr, I32(A, f.offsets), I32(A, f.pair2claim),
                         A + f.has, I32(A, f.rowc), Ba->d_ids, mp, (const int32_t*)Ba->q_lens, A + f.lens_eff);
    else
      hipLaunchKernelGGL(seg_rows_kernel<int64_t>, grid, dim3(256), 0, s, Ba->counts, d.B, d.B1, rg, d.R, d.Mr, I32(A, f.offsets), I32(A, f.pair2claim),
                         A + f.has, I32(A, f.rowc), Ba->d_ids, mp, (const int64_t*)Ba->q_lens, A + f.lens_eff);
    GH_LAUNCH_CHECK();
  } else {
  GH_TRY(gh_seg_offsets(Ba->counts, d.B, I32(A, f.offsets), I32(A, f.pair2claim), d.B1, A + f.has, (void*)s));
  {
    const int nthr = d.Mr > d.B ? d.Mr : d.B;
    hipLaunchKernelGGL(rows_prep_kernel, dim3((nthr + 255) / 256), dim3(256), 0, s, I32(A, f.pair2claim), d.compact ? Ba->rowg : nullptr, d.R,
                       d.Mr, I32(A, f.rowc), Ba->d_ids, d.compact ? nullptr : A + f.maskf_p, Ba->q_lens, Ba->q_lens_kind, A + f.has,
                       A + f.lens_eff, d.B);
    GH_LAUNCH_CHECK();
  }
  }
  // ---- claim branch on the side stream (graph_based_semantic_structure.py:144-155): cell -> masked mean (x has)
  GH_TRY(stream_after(ss, s, ev.ev[0]));
  GH_TRY(cell_fwd(Mo->claim, nullptr, f.q, A, Ba->q_bits, Ba->q_dinv, Ba->q_vals, nullptr, nullptr, 0, 0, Mo->embedding, Ba->q_ids, d.B, d.L,
                  d.D, H, Ba->drop_claim, Ba->seed_claim, nullptr, nullptr, 0.f, 0, ss));
  GH_TRY(gh_masked_mean_fwd(A + f.q.out, Ba->q_ids, A + f.lens_eff, A + f.q_repr, d.B, d.L, H, (void*)ss));
  if (d.cs > 0) {
    unsigned int* cl = clamp_counter();
    GH_REQUIRE(cl, "get_forward: cannot allocate the clamp counter");
    if (Ba->query_sources_i64)
      hipLaunchKernelGGL(left_assemble_fwd_kernel<int64_t>, dim3(d.B), dim3(256), 0, ss, Mo->claim_src_table, (const int64_t*)Ba->query_sources,
                         A + f.q_repr, A + f.has, A + f.new_left, d.cs, H, Mo->claim_src_rows, cl);
    else
      hipLaunchKernelGGL(left_assemble_fwd_kernel<int32_t>, dim3(d.B), dim3(256), 0, ss, Mo->claim_src_table, (const int32_t*)Ba->query_sources,
                         A + f.q_repr, A + f.has, A + f.new_left, d.cs, H, Mo->claim_src_rows, cl);
    GH_LAUNCH_CHECK();
  }
  // the left projections of both attention layers depend on the claim branch only (two_branches_attention.py:137-140:
  // linear1 splits into a left and a right part): they run here, underneath the evidence cells, instead of in front of the
  // two attention GEMMs on the main stream (two split-K launches + their finish kernels, ~45 us of the critical path)
  GH_TRY(att_fwd_impl(A + f.q_repr, d.B, nullptr, nullptr, nullptr, nullptr, nullptr, 0, d.B1, d.R, H, H, H, d.hw, Mo->att_word.w1,
                      Mo->att_word.w2, A + f.uw, nullptr, nullptr, nullptr, nullptr, ss, 1));
  GH_TRY(att_fwd_impl(A + f.new_left, d.B, nullptr, nullptr, nullptr, nullptr, nullptr, 0, d.B, d.n, d.Xl, d.Dre, H, d.he, Mo->att_evd.w1,
                      Mo->att_evd.w2, A + f.ue, nullptr, nullptr, nullptr, nullptr, ss, 1));
  // ---- evidence branch (:107; wrapper.py:165-172): cell -> scorer + top-k -> cell on the refined graph
  const int32_t* goff = d.compact ? Ba->goff : nullptr;
  const int32_t* ids1 = d.compact ? Ba->cids : Ba->d_ids;
  const gh_cell_bf16* c16_1 = d.bf ? &Mo->cell1_16 : nullptr;
  const gh_cell_bf16* c16_2 = d.bf ? &Mo->cell2_16 : nullptr;
  const float* table1 = d.bf ? (const float*)Mo->embedding16 : Mo->embedding;
  uint64_t* keep = reinterpret_cast<uint64_t*>(O + f.keep);
  if (d.fuse_scorer) {      // the scorer's 300 -> 1 projection of the cell output rides in the cell's last epilogue (score_x)
    GH_TRY(cell_fwd(Mo->cell1, c16_1, f.c1, A, Ba->d_bits, Ba->d_dinv, Ba->d_vals, nullptr, goff, d.Mr, d.M1, table1, ids1, d.B1, d.R, d.D, H,
                    Ba->drop_gnn, Ba->seed_cell1, Mo->scorer_w, A + f.score_x, Ba->drop_gnn, Ba->seed_scorer, s, 1));
    GH_TRY(gh_scorer_gsl(Ba->d_bits, Ba->d_dinv, Ba->d_vals, goff, (d.compact && Ba->collapsed) ? 1 : 0, nullptr, A + f.score_x, Mo->scorer_w,
                         Mo->scorer_gate, d.B1, d.R, H, Ba->k_keep, O + f.score, keep, 0.f, 0, (void*)s));
  } else if (d.bf) {        // wide hidden layer, bf16 storage: every column block of the last epilogue reduces its share of the projection
    GH_REQUIRE((H + 127) / 128 <= 8, "get_forward: hidden width %d needs more than the 8 scorer-partial slots", H);      // (before anything is written)
    int parts = 1;          // into a partial of its own; the scorer kernel adds them in block order (no fp32 copy of the cell output at all)
    GH_TRY(cell_fwd(Mo->cell1, c16_1, f.c1, A, Ba->d_bits, Ba->d_dinv, Ba->d_vals, nullptr, goff, d.Mr, d.M1, table1, ids1, d.B1, d.R, d.D, H,
                    Ba->drop_gnn, Ba->seed_cell1, Mo->scorer_w, A + f.score_x, Ba->drop_gnn, Ba->seed_scorer, s, 0, &parts));
    GH_REQUIRE(parts >= 1 && parts <= 8, "get_forward: internal -- %d scorer partials", parts);
    GH_TRY(scorer_gsl_impl(Ba->d_bits, Ba->d_dinv, Ba->d_vals, goff, (d.compact && Ba->collapsed) ? 1 : 0, nullptr, A + f.score_x, parts, d.M1,
                           Mo->scorer_w, Mo->scorer_gate, d.B1, d.R, H, Ba->k_keep, O + f.score, keep, 0.f, 0, s));
  } else {                  // wide hidden layer (h = 768), fp32: a row spans more column blocks than an epilogue can reduce -- the scorer kernel projects
    GH_TRY(cell_fwd(Mo->cell1, c16_1, f.c1, A, Ba->d_bits, Ba->d_dinv, Ba->d_vals, nullptr, goff, d.Mr, d.M1, table1, ids1, d.B1, d.R, d.D, H,
                    Ba->drop_gnn, Ba->seed_cell1, nullptr, nullptr, 0.f, 0, s, 0));
    GH_TRY(gh_scorer_gsl(Ba->d_bits, Ba->d_dinv, Ba->d_vals, goff, (d.compact && Ba->collapsed) ? 1 : 0, A + f.c1.out32, nullptr, Mo->scorer_w,
                         Mo->scorer_gate, d.B1, d.R, H, Ba->k_keep, O + f.score, keep, Ba->drop_gnn, Ba->seed_scorer, (void*)s));
  }
  GH_TRY(cell_fwd(Mo->cell2, c16_2, f.c2, A, Ba->d_bits, Ba->d_dinv, Ba->d_vals, keep, goff, d.Mr, d.Mr, A + f.c1.out, nullptr, d.B1, d.R, H, H,
                  Ba->drop_gnn, Ba->seed_cell2, nullptr, nullptr, 0.f, 0, s));
  GH_TRY(stream_after(s, ss, ev.ev[1]));
  // ---- word-level attention (:173-193): the left input is the claim vector -> ONE u row per claim
  GH_TRY(att_fwd_impl(A + f.q_repr, d.B, I32(A, f.rowc), f.c2.out32 >= 0 ? A + f.c2.out32 : nullptr, d.compact ? Ba->maskf : A + f.maskf_p, goff,
                      d.compact ? Ba->rowg : nullptr, d.Mr, d.B1, d.R, H, H, H, d.hw, Mo->att_word.w1, Mo->att_word.w2, A + f.uw, A + f.tw,
                      A + f.ew, O + f.ww, A + f.avg, s, 2, d.bf ? (const void*)(A + f.c2.out) : nullptr, d.bf ? Mo->att_word_w1_16 : nullptr));
  // ---- evidence-level assembly + attention (:157-171, :195-221)
  GH_TRY(gh_evd_assemble_fwd(A + f.avg, I32(A, f.offsets), Mo->article_src_table, Mo->article_src_rows, d.as > 0 ? Ba->doc_sources : nullptr, Ba->doc_sources_i64,
                             Ba->document, Ba->document_i64, d.B, d.n, d.Xa, d.as, d.R, A + f.right_e, A + f.mask_e, (void*)s));
  GH_TRY(att_fwd_impl(A + f.new_left, d.B, nullptr, A + f.right_e, A + f.mask_e, nullptr, nullptr, 0, d.B, d.n, d.Xl, d.Dre, H, d.he,
                      Mo->att_evd.w1, Mo->att_evd.w2, A + f.ue, A + f.te, A + f.ee, O + f.we, A + f.att_e, s, 2));
  // ---- head (:251-267, :69-74): Linear([claim | attended evidences]) -> Linear, no activation
  if (d.C <= 8) {      // the second layer rides in the first one's finish kernel
    GH_TRY(linear2_fwd(A + f.new_left, d.Xl, A + f.att_e, d.Dre * d.he, Mo->out0_w, Mo->out0_b, A + f.y0, d.B, H, s, Mo->out1_w, Mo->out1_b,
                       O + f.phi, d.C));
  } else {
    GH_TRY(linear2_fwd(A + f.new_left, d.Xl, A + f.att_e, d.Dre * d.he, Mo->out0_w, Mo->out0_b, A + f.y0, d.B, H, s));
    GH_TRY(gh_linear_fwd(A + f.y0, Mo->out1_w, Mo->out1_b, O + f.phi, d.B, H, d.C, (void*)s));
  }
  return 0;
}

extern "C" int gh_get_backward(const gh_get_model* Mo, const gh_get_batch* Ba, const float* A, const float* O, float* Wb, const float* g_phi,
                               const float* g_word_w, const float* g_evd_w, int phase, gh_stream_t stream, gh_stream_t side_stream) {
  Dims d; FwdBuf f; BwdBuf w;
  GH_TRY(layout(Mo, Ba, d, f, w));
  GH_REQUIRE(phase >= 0 && phase <= 2, "get_backward: phase %d not in {0,1,2}", phase);
  GH_REQUIRE(A && O && Wb && g_phi, "get_backward: NULL arena / observables / gradient");
  GH_REQUIRE(Mo->out0_wt && Mo->out1_wt && Mo->att_word.w1t && Mo->att_evd.w1t && Mo->d_out0_w && Mo->d_out0_b && Mo->d_out1_w &&
             Mo->d_out1_b && Mo->att_word.dw1 && Mo->att_word.dw2 && Mo->att_evd.dw1 && Mo->att_evd.dw2,
             "get_backward: transposes / gradient outputs of the attention layers or the head are missing");
  hipStream_t s = (hipStream_t)stream, ss = side_stream ? (hipStream_t)side_stream : s;
  DevEvents ev;
  GH_TRY(get_events(ev));
  const int H = d.H;
  const int32_t* goff = d.compact ? Ba->goff : nullptr;
  const int32_t* ids1 = d.compact ? Ba->cids : Ba->d_ids;
  const uint64_t* keep = reinterpret_cast<const uint64_t*>(O + f.keep);
  // gate heads fused into the producing GEMMs' epilogues: cell scratch order is {dhp, dzp, drp, dxp, da}
  const GateFuse gf2 = {A + f.c2.z, A + f.c2.hh, A + f.c2.xp, Wb + w.sc2[0], Wb + w.sc2[1], Wb + w.sc2[3], d.bf ? 1 : 0};
  const GateFuse gf1 = {A + f.c1.z, A + f.c1.hh, A + f.c1.xp, Wb + w.sc1[0], Wb + w.sc1[1], Wb + w.sc1[3], d.bf ? 1 : 0};
  // bf16 storage: the gate heads' scratch holds bf16; the fused epilogue writes it as such (io bits), which needs the bf16 twins
  // of the word attention's linear1 (the dright product then runs on the bf16-storage kernel).  Without them the gradient
  // between the layers stays an fp32 tensor (g2, dx2) and every cell runs its own gate_bwd_pre pass.
  const bool fuse_gate = !d.bf || (Mo->att_word_w1t_16 != nullptr);
  const void* att_r16 = d.bf ? (const void*)(A + f.c2.out) : nullptr;
  const void* att_w1t16 = d.bf ? Mo->att_word_w1t_16 : nullptr;
  const gh_cell_bf16* c16_1 = d.bf ? &Mo->cell1_16 : nullptr;
  const gh_cell_bf16* c16_2 = d.bf ? &Mo->cell2_16 : nullptr;
  const float* table1 = d.bf ? (const float*)Mo->embedding16 : Mo->embedding;
  if (phase != 2) {
    // ---- head
    static int head_fused = -1;
    if (head_fused < 0) head_fused = measure_env("GH_HEAD_BWD", 1);
    if (head_fused && d.C <= 8 && H % 4 == 0 && Mo->out0_w && Mo->out1_w && head_bwd_lds(d.B, H, d.C, d.E, Mo->out0_w) > 0) {
      // both layers' input gradients and the second layer's weight gradients in one launch (head_bwd_kernel)
      GH_TRY(launch_head_bwd(g_phi, A + f.y0, Mo->out1_w, Mo->out0_w, d.B, H, d.C, d.Xl, d.E, Wb + w.d_y0, Mo->d_out1_w, Mo->d_out1_b,
                             Wb + w.d_new_left, 0, Wb + w.d_att_e, s));
    } else {
      GH_TRY(gh_linear_bwd(A + f.y0, Mo->out1_wt, Mo->out1_w, g_phi, d.B, H, d.C, Wb + w.d_y0, Mo->d_out1_w, Mo->d_out1_b, (void*)s));
      GH_TRY(linear2_bwd(A + f.new_left, d.Xl, A + f.att_e, d.Dre * d.he, Mo->out0_wt, Wb + w.d_y0, d.B, H, Wb + w.d_new_left, 0,
                         Wb + w.d_att_e, nullptr, nullptr, s));
    }
    GH_TRY(stream_after(ss, s, ev.ev[2]));          // few-row weight gradients leave the critical path
    GH_TRY(linear2_bwd(A + f.new_left, d.Xl, A + f.att_e, d.Dre * d.he, Mo->out0_wt, Wb + w.d_y0, d.B, H, nullptr, 0, nullptr,
                       Mo->d_out0_w, Mo->d_out0_b, ss));
    // ---- evidence-level attention; its left gradient adds to the head's
    //      (the left gradients -- d_new_left here, d_q below -- feed the claim branch only: their GEMMs run with the linear1 weight
    //       gradients on the side stream, `dleft_late`, and so does the split of d_new_left into d_q + claim-source table gradient)
    GH_TRY(att_bwd_impl(A + f.new_left, A + f.right_e, nullptr, 0, d.B, d.n, d.Xl, d.Dre, H, d.he, Mo->att_evd.w1t, Mo->att_evd.w2, A + f.te,
                        O + f.we, Wb + w.d_att_e, g_evd_w, Wb + w.de_e, Wb + w.dpre_e, Wb + w.du_e, Wb + w.d_new_left, Wb + w.dright_e,
                        nullptr, Mo->att_evd.dw2, nullptr, d.B, nullptr, 1, s, nullptr, (d.Dre + 511) / 512 <= 16 ? Wb + w.dw_e : nullptr, nullptr, 1,
                        Wb + w.dw2p_e));
    GH_TRY(stream_after(ss, s, ev.ev[3]));
    GH_TRY(att_bwd_impl(A + f.new_left, A + f.right_e, nullptr, 0, d.B, d.n, d.Xl, d.Dre, H, d.he, Mo->att_evd.w1t, Mo->att_evd.w2, A + f.te,
                        O + f.we, Wb + w.d_att_e, g_evd_w, Wb + w.de_e, Wb + w.dpre_e, Wb + w.du_e, Wb + w.d_new_left, nullptr, Mo->att_evd.dw1,
                        Mo->att_evd.dw2, nullptr, d.B, nullptr, 1, ss, nullptr, nullptr, nullptr, 1, Wb + w.dw2p_e));
    if (d.cs > 0) {
      if (Ba->query_sources_i64)
        hipLaunchKernelGGL(left_assemble_bwd_kernel<int64_t>, dim3(1), dim3(256), 0, ss, Wb + w.d_new_left, (const int64_t*)Ba->query_sources,
                           A + f.has, Wb + w.d_q, Mo->d_claim_src_table, d.B, d.cs, H, Mo->claim_src_rows);
      else
        hipLaunchKernelGGL(left_assemble_bwd_kernel<int32_t>, dim3(1), dim3(256), 0, ss, Wb + w.d_new_left, (const int32_t*)Ba->query_sources,
                           A + f.has, Wb + w.d_q, Mo->d_claim_src_table, d.B, d.cs, H, Mo->claim_src_rows);
      GH_LAUNCH_CHECK();
    }
    // ---- evidence-level assembly: d_avg (rows of claims with more than n_max evidences stay zero), article-source table
    // (rows of claims with more than n_max evidences have no slot in the assembly: its backward writes their zeros itself -- the
    //  claim's last slot does -- so no fill of d_avg runs in front of it; ADVICE r3's uninitialised-memory case stays covered)
    GH_TRY(gh_evd_assemble_bwd(Wb + w.dright_e, I32(A, f.offsets), d.as > 0 ? Ba->doc_sources : nullptr, Ba->doc_sources_i64, Mo->article_src_rows, d.B, d.n, d.Xa,
                               d.as, Wb + w.d_avg, d.as > 0 ? Mo->d_article_src_table : nullptr, (void*)s));
    // ---- word-level attention: per-pair du summed per claim; d(claim vector) accumulates on top of the head / evidence part.
    //      Its dright GEMM produces the gradient of the second evidence cell's output and nothing else reads it: the GEMM's
    //      epilogue applies that cell's gate head (gf2); likewise the second cell's dX GEMM feeds the first cell's (gf1).
    GH_TRY(att_bwd_impl(A + f.q_repr, f.c2.out32 >= 0 ? A + f.c2.out32 : nullptr, goff, d.Mr, d.B1, d.R, H, H, H, d.hw, Mo->att_word.w1t, Mo->att_word.w2, A + f.tw,
                        O + f.ww, Wb + w.d_avg, g_word_w, Wb + w.de_w, Wb + w.dpre_w, Wb + w.du_w, Wb + w.d_q, Wb + w.g2, nullptr,
                        Mo->att_word.dw2, I32(A, f.offsets), d.B, Wb + w.du_c, 1, s, d.compact ? Ba->rowg : nullptr, Wb + w.dw_w,
                        fuse_gate ? &gf2 : nullptr, 1, Wb + w.dw2p_w, att_r16, att_w1t16));
    // ---- side stream: linear1's weight gradient of the word attention, then the claim branch's backward -- underneath
    //      the evidence cells' chain on the main stream
    GH_TRY(stream_after(ss, s, ev.ev[4]));
    GH_TRY(att_bwd_impl(A + f.q_repr, f.c2.out32 >= 0 ? A + f.c2.out32 : nullptr, goff, d.Mr, d.B1, d.R, H, H, H, d.hw, Mo->att_word.w1t, Mo->att_word.w2, A + f.tw,
                        O + f.ww, Wb + w.d_avg, g_word_w, Wb + w.de_w, Wb + w.dpre_w, Wb + w.du_w, Wb + w.d_q, nullptr, Mo->att_word.dw1,
                        Mo->att_word.dw2, I32(A, f.offsets), d.B, Wb + w.du_c, 1, ss, nullptr, nullptr, nullptr, 1, Wb + w.dw2p_w, att_r16,
                        att_w1t16));
    GH_TRY(gh_masked_mean_bwd(Wb + w.d_q, Ba->q_ids, A + f.lens_eff, Wb + w.d_qhid, d.B, d.L, H, (void*)ss));
    GH_TRY(cell_bwd(Mo->claim, nullptr, f.q, A, Ba->q_bits, Ba->q_dinv, Ba->q_vals, nullptr, nullptr, 0, Mo->embedding, Ba->q_ids, d.B, d.L, d.D, H,
                    Wb + w.d_qhid, Wb, w.qs, nullptr, Ba->drop_claim, Ba->seed_claim, ss));
    // ---- second evidence cell.  (cell_bwd_impl can put the weight-gradient GEMMs on the side stream -- measured on the
    //      bench step: 6.31 -> 6.23 ms, two MFMA-bound streams mostly slow each other down, while every kernel's wall time
    //      and with it the per-kernel roofline figures inflate by 20-30 %.  Not used: one stream, honest kernel times.)
    if (fuse_gate)
      GH_TRY(cell_bwd(Mo->cell2, c16_2, f.c2, A, Ba->d_bits, Ba->d_dinv, Ba->d_vals, keep, goff, d.Mr, A + f.c1.out, nullptr, d.B1, d.R, H, H,
                      Wb + w.g2, Wb, w.sc2, nullptr, Ba->drop_gnn, Ba->seed_cell2, s, 1, &gf1));
    else
      GH_TRY(cell_bwd(Mo->cell2, c16_2, f.c2, A, Ba->d_bits, Ba->d_dinv, Ba->d_vals, keep, goff, d.Mr, A + f.c1.out, nullptr, d.B1, d.R, H, H,
                      Wb + w.g2, Wb, w.sc2, Wb + w.dx2, Ba->drop_gnn, Ba->seed_cell2, s, 0, nullptr));
  }
  if (phase != 1) {
    if (fuse_gate)
      GH_TRY(cell_bwd(Mo->cell1, c16_1, f.c1, A, Ba->d_bits, Ba->d_dinv, Ba->d_vals, nullptr, goff, d.Mr, table1, ids1, d.B1, d.R, d.D, H,
                      nullptr, Wb, w.sc1, nullptr, Ba->drop_gnn, Ba->seed_cell1, s, 1, nullptr));
    else
      GH_TRY(cell_bwd(Mo->cell1, c16_1, f.c1, A, Ba->d_bits, Ba->d_dinv, Ba->d_vals, nullptr, goff, d.Mr, table1, ids1, d.B1, d.R, d.D, H,
                      Wb + w.dx2, Wb, w.sc1, nullptr, Ba->drop_gnn, Ba->seed_cell1, s, 0, nullptr));
    GH_TRY(stream_after(s, ss, ev.ev[5]));
  }
  return 0;
}

extern "C" int gh_cross_entropy(const float* phi, const int64_t* labels, int b, int c, float* loss, float* dphi, gh_stream_t stream) {
  GH_REQUIRE(b > 0 && c > 0 && phi && labels && loss && dphi, "cross_entropy: bad arguments");
  unsigned int* cl = clamp_counter();
  GH_REQUIRE(cl, "cross_entropy: cannot allocate the clamp counter");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, phi, labels, b, c, loss, dphi, cl);
  GH_LAUNCH_CHECK();
  return 0;
}

extern "C" int gh_get_prepare(const int32_t* claim_tokens, const int32_t* claim_len, int b, int l,
                              const int32_t* evd_tokens, const int32_t* evd_len, int b1, int r, int window,
                              int32_t* q_ids, int32_t* q_n, uint64_t* q_bits, float* q_dinv,
                              int32_t* d_ids, int32_t* d_n, uint64_t* d_bits, float* d_dinv,
                              int m_real, int32_t* goff, int32_t* rowg, int32_t* src, int32_t* cids, float* maskf,
                              const int64_t* slot, int32_t* document, gh_stream_t stream) {
  // two launches: both sides' graphs, then the node-compact plan with the document scatter on its way (five launches before)
  if (b > 0 && b1 > 0) {
    GH_TRY(launch_graph_build2(claim_tokens, claim_len, b, l, q_ids, q_n, q_bits, q_dinv, evd_tokens, evd_len, b1, r, d_ids, d_n, d_bits,
                               d_dinv, window, (hipStream_t)stream));
  } else {
    GH_TRY(gh_graph_build(claim_tokens, claim_len, b, l, window, q_ids, q_n, q_bits, q_dinv, stream));
    GH_TRY(gh_graph_build(evd_tokens, evd_len, b1, r, window, d_ids, d_n, d_bits, d_dinv, stream));
  }
  bool scattered = false;
  if (m_real >= 0 && b1 > 0) {
    GH_TRY(launch_ragged_plan(d_n, d_ids, b1, r, goff, rowg, src, cids, maskf, slot, document, (hipStream_t)stream, &scattered));
  }
  if (slot && document && b1 > 0 && !scattered) {
    hipLaunchKernelGGL(document_scatter_kernel, dim3(b1), dim3(256), 0, (hipStream_t)stream, d_ids, slot, document, r);
    GH_LAUNCH_CHECK();
  }
  return 0;
}
