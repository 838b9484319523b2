// Graph-side kernels: word-graph build, adjacency packing, bitmask aggregation (spmm),
// fused word-scorer + GSL top-k.  All HBM-bound integer/bit work plus one streaming FMA pass;
// no MFMA here by design.
#include "../../include/get_hip.h"
#include "common.h"
#include "gemm.hip.h"
#include <stdlib.h>

namespace gh {

constexpr int MAX_R = 256;          // padded graph size supported by the one-workgroup-per-graph kernels
constexpr int MAX_W = MAX_R / 64;

// ------------------------------------------------------------------------------------------------
// a1  graph build (interactions.py:334-351).  One workgroup per text.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void graph_build_body(const int g, const int32_t* __restrict__ tokens, const int32_t* __restrict__ lengths, int R,
                                                 int window, int32_t* __restrict__ node_ids, int32_t* __restrict__ n_nodes,
                                                 uint64_t* __restrict__ bits, float* __restrict__ dinv) {
  __shared__ int tok[MAX_R];
  __shared__ int first[MAX_R];       // position of the first occurrence of tok[i]
  __shared__ int node_of[MAX_R];     // node index of position i
  __shared__ unsigned long long rows[MAX_R * MAX_W];
  __shared__ int total;
  const int tid = threadIdx.x;
  const int W = (R + 63) / 64;
  int len = lengths[g];
  len = len < 0 ? 0 : (len > R ? R : len);
  for (int i = tid; i < R; i += blockDim.x) tok[i] = (i < len) ? tokens[(size_t)g * R + i] : 0;
  for (int i = tid; i < R * W; i += blockDim.x) rows[i] = 0ull;
  __syncthreads();
  // first occurrence (words_list.sort(key=raw_text.index), :335-336)
  for (int i = tid; i < len; i += blockDim.x) {
    int f = i;
    const int t = tok[i];
    for (int j = 0; j < i; ++j)
      if (tok[j] == t) { f = j; break; }
    first[i] = f;
  }
  __syncthreads();
  // node index = number of first-occurrence positions before first[i]
  for (int i = tid; i < len; i += blockDim.x) {
    const int f = first[i];
    int c = 0;
    for (int j = 0; j < f; ++j) c += (first[j] == j);
    node_of[i] = c;
  }
  if (tid == 0) {
    int c = 0;
    for (int j = 0; j < len; ++j) c += (first[j] == j);
    total = c;
  }
  __syncthreads();
  // node list, zero padded (:349)
  for (int i = tid; i < R; i += blockDim.x) node_ids[(size_t)g * R + i] = 0;
  __syncthreads();
  for (int i = tid; i < len; i += blockDim.x)
    if (first[i] == i) node_ids[(size_t)g * R + node_of[i]] = tok[i];
  // window links over POSITIONS, accumulated on nodes (:342-344)
  for (int i = tid; i < len; i += blockDim.x) {
    const int ni = node_of[i];
    const int lo = max(i - window + 1, 0), hi = min(i + window, len);
    for (int j = lo; j < hi; ++j) {
      const int nj = node_of[j];
      atomicOr(&rows[ni * W + (nj >> 6)], 1ull << (nj & 63));
    }
  }
  __syncthreads();
  for (int i = tid; i < R; i += blockDim.x) {
    int deg = 0;
    for (int w = 0; w < W; ++w) {
      const unsigned long long m = rows[i * W + w];
      bits[((size_t)g * R + i) * W + w] = m;
      deg += __popcll(m);
    }
    // D^-1/2 with zero-degree rows -> 0 (interactions.py:14-16)
    dinv[(size_t)g * R + i] = deg > 0 ? (float)(1.0 / sqrt((double)deg)) : 0.f;
  }
  if (tid == 0) n_nodes[g] = total;
}
__global__ void __launch_bounds__(256)
graph_build_kernel(const int32_t* __restrict__ tokens, const int32_t* __restrict__ lengths, int R, int window,
                   int32_t* __restrict__ node_ids, int32_t* __restrict__ n_nodes, uint64_t* __restrict__ bits,
                   float* __restrict__ dinv) {
  graph_build_body(blockIdx.x, tokens, lengths, R, window, node_ids, n_nodes, bits, dinv);
}
// both sides of a batch in one launch (gh_get_prepare): workgroups [0, na) build the claims, the rest the evidences
struct GraphSide { const int32_t* tokens; const int32_t* lengths; int R; int32_t* node_ids; int32_t* n_nodes; uint64_t* bits; float* dinv; };
__global__ void __launch_bounds__(256)
graph_build2_kernel(const GraphSide a, const GraphSide b, int na, int window) {
  const bool first = (int)blockIdx.x < na;      // workgroup-uniform
  const GraphSide& s = first ? a : b;
  graph_build_body(first ? blockIdx.x : blockIdx.x - na, s.tokens, s.lengths, s.R, window, s.node_ids, s.n_nodes, s.bits, s.dinv);
}

// ------------------------------------------------------------------------------------------------
// dense (N,R,R) adjacency -> packed bits + fp32 values.  One wave per row, ballot builds the word.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
adj_pack_kernel(const T* __restrict__ adj, int R, uint64_t* __restrict__ bits, float* __restrict__ vals) {
  const int g = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int W = (R + 63) / 64;
  for (int i = wave; i < R; i += 4) {
    const size_t rb = ((size_t)g * R + i) * R;
    for (int w = 0; w < W; ++w) {
      const int j = w * 64 + lane;
      float v = 0.f, vt = 0.f;
      if (j < R) {
        v = (float)adj[rb + j];
        vals[rb + j] = v;
        vt = (float)adj[((size_t)g * R + j) * R + i];
      }
      // the bit pattern is SYMMETRISED (A[i][j] != 0 or A[j][i] != 0): the transposed aggregation of the backward walks
      // row i's bits and reads vals[j][i], so an entry present only in A^T must have its bit too; the extra entries
      // carry the value 0 in `vals` and change no result
      const unsigned long long m = __ballot(v != 0.f || vt != 0.f);
      if (lane == 0) bits[((size_t)g * R + i) * W + w] = m;
    }
  }
}

// The reference fitter's de-padding (char_man_fitter_query_repr1.py:204-250: a Python loop over the claims slicing
// `[:evd_count]` out of the padded (B, n_max, R) ids and (B, n_max, R, R) float64 adjacency, two host syncs per claim) plus the
// packing of the adjacency and the node counts the node-compact plan needs, in ONE pass over the handed-over tensors:
// workgroup (claim c, slot j) with j < counts[c] is pair p = sum(counts[:c]) + j; it narrows the ids, counts the real nodes
// (id >= 1), packs its R x R block (as adj_pack_kernel) and checks what the node-compact layout assumes (ids prefix-shaped,
// no edge on a padding node).  It also recognises the adjacency convert_text produces (interactions.py:11-18: D^-1/2 A D^-1/2
// of a binary graph, A[i][j] = dinv[i] dinv[j] with dinv = 1 / sqrt(row degree)): such a block is fully described by its bit
// rows + dinv (the kernels' "normalised" mode, 2 KB instead of 40 KB per graph); only for other blocks, or when `force_vals`
// says that some graph of the batch needs them, the dense fp32 values are written.
// stats = {pairs, real nodes, pairs that violate the layout assumption, pairs that are NOT normalised graphs, reserved}.
template <typename TI>
__global__ void __launch_bounds__(256)
ref_depad_kernel(const int64_t* __restrict__ counts, int B, int n_max, int R, const TI* __restrict__ ids, const double* __restrict__ adj,
                 int32_t* __restrict__ d_ids, uint64_t* __restrict__ bits, float* __restrict__ vals, float* __restrict__ dinv,
                 int32_t* __restrict__ n_nodes, unsigned long long* __restrict__ stats, int force_vals) {
  const int c = blockIdx.x / n_max, j = blockIdx.x % n_max;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int red[4];
  __shared__ int s_real[256];
  __shared__ float s_dinv[256];
  __shared__ int s_bad, s_weighted;
  const long long cnt_c = counts[c];
  const int cc = cnt_c < 0 ? 0 : (cnt_c > n_max ? n_max : (int)cnt_c);
  if (j >= cc) return;
  // pair index: clamped counts of the claims before c (B is a few hundred at most)
  int part = 0;
  for (int i = tid; i < c; i += 256) { const long long v = counts[i]; part += v < 0 ? 0 : (v > n_max ? n_max : (int)v); }
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) red[wave] = part;
  if (tid == 0) { s_bad = 0; s_weighted = 0; }
  __syncthreads();
  const int p = red[0] + red[1] + red[2] + red[3] + j;
  const size_t slot = (size_t)c * n_max + j;
  // ids -> int32, real-node flags, node count
  int real = 0;
  if (tid < R) {
    const long long id = (long long)ids[slot * R + tid];
    d_ids[(size_t)p * R + tid] = (int32_t)id;
    real = id >= 1 ? 1 : 0;
  }
  s_real[tid] = real;
  int nn = real;
  for (int o = 32; o > 0; o >>= 1) nn += __shfl_xor(nn, o);
  __syncthreads();                      // (red is re-used: every thread has read it)
  if (lane == 0) red[wave] = nn;
  __syncthreads();
  nn = red[0] + red[1] + red[2] + red[3];
  int bad = (tid < R && real != (tid < nn ? 1 : 0)) ? 1 : 0;
  // pass 1 over the block: bit rows (union of A's and A^T's patterns) and row degrees -> dinv as gh_graph_build computes it
  const int W = (R + 63) / 64;
  const double* ab = adj + slot * (size_t)R * R;
  for (int i = wave; i < R; i += 4) {
    int deg = 0;
    for (int w = 0; w < W; ++w) {
      const int jj = w * 64 + lane;
      double v = 0.0, vt = 0.0;
      if (jj < R) { v = ab[(size_t)i * R + jj]; vt = ab[(size_t)jj * R + i]; }
      const unsigned long long m = __ballot((float)v != 0.f || (float)vt != 0.f);
      if (lane == 0) bits[((size_t)p * R + i) * W + w] = m;
      deg += __popcll(m);
    }
    if (deg > 0 && !s_real[i]) bad = 1;     // a padding node with edges: the node-compact layout would drop them
    if (lane == 0) s_dinv[i] = deg > 0 ? (float)(1.0 / sqrt((double)deg)) : 0.f;
  }
  if (bad) s_bad = 1;
  __syncthreads();
  if (tid < R) dinv[(size_t)p * R + tid] = s_dinv[tid];
  // pass 2 (the block is in L2 now): is every entry dinv[i] dinv[j] (to fp32 rounding of the fitter's float64 product)?
  int weighted = 0;
  for (int i = wave; i < R; i += 4) {
    const float di = s_dinv[i];
    for (int w = 0; w < W; ++w) {
      const int jj = w * 64 + lane;
      if (jj < R) {
        const float v = (float)ab[(size_t)i * R + jj];
        const float e = di * s_dinv[jj];
        // (a pattern entry whose own value is 0 -- present on the transposed side only -- is not a normalised graph either)
        const bool on = v != 0.f || (float)ab[(size_t)jj * R + i] != 0.f;
        if (on && !(fabsf(v - e) <= 4e-7f * e)) weighted = 1;
      }
    }
  }
  if (weighted) s_weighted = 1;
  __syncthreads();
  if (s_weighted || force_vals) {
    for (int i = wave; i < R; i += 4)
      for (int jj = lane; jj < R; jj += 64) vals[((size_t)p * R + i) * R + jj] = (float)ab[(size_t)i * R + jj];
  }
  if (tid == 0) {
    n_nodes[p] = nn;
    atomicAdd(&stats[0], 1ull);
    atomicAdd(&stats[1], (unsigned long long)nn);
    if (s_bad) atomicAdd(&stats[2], 1ull);
    if (s_weighted) atomicAdd(&stats[3], 1ull);
  }
}

__global__ void __launch_bounds__(256)
adj_unpack_kernel(const uint64_t* __restrict__ bits, const float* __restrict__ dinv, const float* __restrict__ vals,
                  const uint64_t* __restrict__ keep, int R, float* __restrict__ adj) {
  const int g = blockIdx.x;
  const int W = (R + 63) / 64;
  for (int it = threadIdx.x; it < R * R; it += blockDim.x) {
    const int i = it / R, j = it % R;
    bool on = (bits[((size_t)g * R + i) * W + (j >> 6)] >> (j & 63)) & 1ull;
    if (on && keep) {
      const bool ki = (keep[(size_t)g * W + (i >> 6)] >> (i & 63)) & 1ull;
      const bool kj = (keep[(size_t)g * W + (j >> 6)] >> (j & 63)) & 1ull;
      on = ki || kj;
    }
    float v = 0.f;
    if (on) v = vals ? vals[((size_t)g * R + i) * R + j] : dinv[(size_t)g * R + i] * dinv[(size_t)g * R + j];
    adj[((size_t)g * R + i) * R + j] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// aggregation  y[g][i][:] (+)= sum_{j in N(i)} w_ij x[g][j][:]       (wrapper.py:192)
// grid (graph, column slab).  The slab of x is staged in LDS once (coalesced 16 B/lane reads of the
// feature rows), the refined neighbour bit-rows and dinv sit beside it, and every output float4
// walks its row's set bits with ctz -- each feature row leaves HBM exactly once per slab.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 bf4_to_f4(uint2 u) {
  return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                     __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
}
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint2 f4_to_bf4(float4 v) { return make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)); }

// BF (with V = 4): x and y hold bf16 (the bf16 storage pipeline); the slab is widened to fp32 when it is staged
template <int V, bool BF = false>   // V = 4: float4 columns, V = 1: scalar columns
__global__ void __launch_bounds__(256)
spmm_kernel(const uint64_t* __restrict__ bits, const float* __restrict__ dinv, const float* __restrict__ vals,
            const uint64_t* __restrict__ keep, const int32_t* __restrict__ goff, const float* __restrict__ x,
            float* __restrict__ y, int R, int H, int slab, int transpose, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int W = (R + 63) / 64;
  float* xs = reinterpret_cast<float*>(dsm);                                  // [R][slab*V]
  const size_t xs_floats = (((size_t)R * slab * V) + 3) & ~(size_t)3;           // keep rb 16 B aligned
  unsigned long long* rb = reinterpret_cast<unsigned long long*>(xs + xs_floats);              // [R][W]
  float* dv = reinterpret_cast<float*>(rb + (size_t)R * W);                    // [R]
  const int g = blockIdx.x, tid = threadIdx.x;
  const int c0 = blockIdx.y * slab;                 // first column (in units of V floats)
  const int HV = H / V;
  const int ncol = min(slab, HV - c0);
  // node-compact layout: graph g owns feature rows [goff[g], goff[g+1]) -- its real nodes only; bit rows, dinv and
  // vals keep their padded [g][R] indexing (node i of graph g is feature row goff[g] + i)
  const int row0 = goff ? goff[g] : g * R;
  const int NR = goff ? goff[g + 1] - row0 : R;
  if (NR <= 0) return;
  const float* xg = x + (size_t)row0 * H;
  // Stage everything with ONE memory round trip: the slab (up to SL 16-byte loads per thread from a clamped
  // index -- unconditional, so no branch or per-load s_waitcnt), this thread's bit-row word and dinv entry
  // are all issued back to back; sched_barrier keeps them ahead of the first LDS write (the scheduler otherwise
  // pairs each load with its store: one HBM latency per element).  Out-of-range slots rewrite the last
  // element with its own value, which keeps the stores unconditional too.
  {
    constexpr int SL = 12;
    const int total = NR * ncol;
    // bit rows / dinv first (R*W <= 1024 and R <= 256 -> at most 4 + 1 per thread); consumed after the slab
    unsigned long long mw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) mw[u] = bits[(size_t)g * R * W + min(tid + u * 256, R * W - 1)];
    const float dvv = vals ? 0.f : dinv[(size_t)g * R + min(tid, R - 1)];
    for (int base = tid; base < total; base += SL * 256) {
      float4 tmp[SL];
#pragma unroll
      for (int k = 0; k < SL; ++k) {
        const int it = min(base + k * 256, total - 1);
        const int i = it / ncol, c = it % ncol;
        if (BF) tmp[k] = bf4_to_f4(reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x) + ((size_t)row0 + i) * H)[c0 + c]);
        else if (V == 4) tmp[k] = reinterpret_cast<const float4*>(xg + (size_t)i * H)[c0 + c];
        else tmp[k].x = xg[(size_t)i * H + c0 + c];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < SL; ++k) {
        const int it = min(base + k * 256, total - 1);
        const int i = it / ncol, c = it % ncol;
        if (V == 4) reinterpret_cast<float4*>(xs)[i * slab + c] = tmp[k];
        else xs[i * slab + c] = tmp[k].x;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int it = min(tid + u * 256, R * W - 1);
      unsigned long long m = mw[u];
      if (keep) {
        const int i = it / W, w = it % W;
        const bool ki = (keep[(size_t)g * W + (i >> 6)] >> (i & 63)) & 1ull;
        if (!ki) m &= keep[(size_t)g * W + w];          // edge survives iff keep(i) || keep(j)
      }
      rb[it] = m;
    }
    if (!vals) dv[min(tid, R - 1)] = dvv;
  }
  __syncthreads();
  const float* vg = vals ? vals + (size_t)g * R * R : nullptr;
  float* yg = y + (size_t)row0 * H;
  for (int it = tid; it < NR * ncol; it += 256) {
    const int i = it / ncol, c = it % ncol;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float di = vals ? 0.f : dv[i];
    for (int w = 0; w < W; ++w) {
      unsigned long long m = rb[i * W + w];
      while (m) {
        const int j = (w << 6) + __builtin_ctzll(m);
        m &= m - 1;
        const float wt = vg ? (transpose ? vg[(size_t)j * R + i] : vg[(size_t)i * R + j]) : di * dv[j];
        if (V == 4) {
          const float4 xv = reinterpret_cast<const float4*>(xs)[j * slab + c];
          acc.x += wt * xv.x; acc.y += wt * xv.y; acc.z += wt * xv.z; acc.w += wt * xv.w;
        } else {
          acc.x += wt * xs[j * slab + c];
        }
      }
    }
    if (BF) {
      uint2* o = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(y) + ((size_t)row0 + i) * H) + c0 + c;
      if (accumulate) { const float4 p = bf4_to_f4(*o); acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
      *o = f4_to_bf4(acc);
    } else if (V == 4) {
      float4* o = reinterpret_cast<float4*>(yg + (size_t)i * H) + c0 + c;
      if (accumulate) { const float4 p = *o; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
      *o = acc;
    } else {
      float* o = yg + (size_t)i * H + c0 + c;
      *o = accumulate ? *o + acc.x : acc.x;
    }
  }
}

// LDS-free variant for float4-shaped rows: every thread owns output float4s (row i, column c) of one
// graph and gathers its neighbours' float4s straight from L1/L2 (neighbours of node i are almost always
// the adjacent nodes, so the gathers hit lines its wave neighbours just touched).  No staging phase and
// no barrier: ~8 independent 16-byte loads in flight per thread, 8 waves per SIMD.
__global__ void __launch_bounds__(256)
spmm_gather_kernel(const uint64_t* __restrict__ bits, const float* __restrict__ dinv, const float* __restrict__ vals,
                   const uint64_t* __restrict__ keep, const float* __restrict__ x, float* __restrict__ y, int R, int H,
                   int transpose, int accumulate, int blocks_per_graph) {
  const int g = blockIdx.x / blocks_per_graph, part = blockIdx.x % blocks_per_graph;
  const int W = (R + 63) / 64, H4 = H / 4;
  const float4* xg = reinterpret_cast<const float4*>(x + (size_t)g * R * H);
  float4* yg = reinterpret_cast<float4*>(y + (size_t)g * R * H);
  const uint64_t* bg = bits + (size_t)g * R * W;
  const uint64_t* kg = keep ? keep + (size_t)g * W : nullptr;
  const float* dg = dinv ? dinv + (size_t)g * R : nullptr;
  const float* vg = vals ? vals + (size_t)g * R * R : nullptr;
  const int total = R * H4;
  for (int it = part * 256 + threadIdx.x; it < total; it += blocks_per_graph * 256) {
    const int i = it / H4, c = it % H4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float di = dg ? dg[i] : 0.f;
    bool ki = true;
    if (kg) ki = (kg[i >> 6] >> (i & 63)) & 1ull;
    for (int w = 0; w < W; ++w) {
      unsigned long long m = bg[(size_t)i * W + w];
      if (kg && !ki) m &= kg[w];
      // neighbours in batches of 8: collect the indices first (pure bit work), then issue all 16-byte
      // gathers back to back so their latencies overlap instead of chaining
      while (m) {
        int nb[8];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const bool on = m != 0ull;
          nb[q] = on ? (w << 6) + __builtin_ctzll(m) : 0;
          cnt += on;
          m &= m - 1;                       // m == 0 stays 0
        }
        float4 xv[8];
        float wt[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q < cnt) {
            xv[q] = xg[(size_t)nb[q] * H4 + c];
            wt[q] = vg ? (transpose ? vg[(size_t)nb[q] * R + i] : vg[(size_t)i * R + nb[q]]) : di * dg[nb[q]];
          }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q < cnt) {
            acc.x += wt[q] * xv[q].x; acc.y += wt[q] * xv[q].y; acc.z += wt[q] * xv[q].z; acc.w += wt[q] * xv[q].w;
          }
      }
    }
    float4* o = yg + (size_t)i * H4 + c;
    if (accumulate) { const float4 p = *o; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
    *o = acc;
  }
}

// Wave-per-row variant: the neighbour bit rows are wave-uniform (scalar registers, SALU bit walk), every
// lane owns one or two float4 columns of the output row, neighbours are consumed four at a time so that
// up to eight 16-byte gathers are in flight per lane with no divergence at all.
template <int RPW>
__global__ void __launch_bounds__(256)
spmm_wave_kernel(const uint64_t* __restrict__ bits, const float* __restrict__ dinv, const float* __restrict__ vals,
                 const uint64_t* __restrict__ keep, const float* __restrict__ x, float* __restrict__ y, int R, int H,
                 int transpose, int accumulate, int waves_per_graph) {
  const int lane = threadIdx.x & 63;
  const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int g = gw / waves_per_graph, r0 = (gw % waves_per_graph) * RPW;
  const int W = (R + 63) / 64, H4 = H / 4;
  const float4* xg = reinterpret_cast<const float4*>(x + (size_t)g * R * H);
  float4* yg = reinterpret_cast<float4*>(y + (size_t)g * R * H);
  const uint64_t* bg = bits + (size_t)g * R * W;
  const uint64_t* kg = keep ? keep + (size_t)g * W : nullptr;
  const float* dg = dinv ? dinv + (size_t)g * R : nullptr;
  const float* vg = vals ? vals + (size_t)g * R * R : nullptr;
  const bool two = lane + 64 < H4;
  for (int i = r0; i < min(R, r0 + RPW); ++i) {
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    const float di = dg ? dg[i] : 0.f;
    bool ki = true;
    if (kg) ki = (kg[i >> 6] >> (i & 63)) & 1ull;
    for (int w = 0; w < W; ++w) {
      unsigned long long m = bg[(size_t)i * W + w];
      if (kg && !ki) m &= kg[w];
      while (m) {
        int nb[4];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool on = m != 0ull;
          nb[q] = on ? (w << 6) + __builtin_ctzll(m) : 0;
          cnt += on;
          m &= m - 1;
        }
        float4 v0[4], v1[4];
        float wt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < cnt) {
            if (lane < H4) v0[q] = xg[(size_t)nb[q] * H4 + lane];
            if (two) v1[q] = xg[(size_t)nb[q] * H4 + lane + 64];
            wt[q] = vg ? (transpose ? vg[(size_t)nb[q] * R + i] : vg[(size_t)i * R + nb[q]]) : di * dg[nb[q]];
          }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < cnt) {
            a0.x += wt[q] * v0[q].x; a0.y += wt[q] * v0[q].y; a0.z += wt[q] * v0[q].z; a0.w += wt[q] * v0[q].w;
            a1.x += wt[q] * v1[q].x; a1.y += wt[q] * v1[q].y; a1.z += wt[q] * v1[q].z; a1.w += wt[q] * v1[q].w;
          }
      }
    }
    if (lane < H4) {
      float4* o = yg + (size_t)i * H4 + lane;
      if (accumulate) { const float4 p = *o; a0.x += p.x; a0.y += p.y; a0.z += p.z; a0.w += p.w; }
      *o = a0;
    }
    if (two) {
      float4* o = yg + (size_t)i * H4 + lane + 64;
      if (accumulate) { const float4 p = *o; a1.x += p.x; a1.y += p.y; a1.z += p.z; a1.w += p.w; }
      *o = a1;
    }
  }
}

// Edge-list variant of the slab kernel (the default for float4-shaped rows).  The bit-walk version above spends ~25
// VALU instructions per (output float4, neighbour) -- 64-bit ctz / clear-lowest-bit / weight product per lane -- and was
// VALU-issue-bound (1.4 G lane-instructions per launch at the bench shape = 35 us of a 47 us launch).  Here the
// refined bit rows of the graph are expanded ONCE per workgroup into an LDS edge list {j, w_ij} (row-start offsets by a
// workgroup scan of the popcounts), and every thread owns CPT float4 columns of one row, so the inner loop per
// neighbour is one 8-byte list read, one address mad and CPT x (ds_read_b128 + 4 FMA).  Graphs whose edge count
// exceeds the list capacity (dense hand-overs) fall back to the bit walk inside the same kernel.
#ifdef GH_MEASURE
__device__ unsigned g_spmm_phase[8192 * 8];     // tool build: s_memtime ticks per phase and workgroup (thread 0); ~0.5 ns per tick on the box (calibrated on the kernel time)
#define SPMM_T(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (blockIdx.x < 8192) g_spmm_phase[blockIdx.x * 8 + i] = (unsigned)(t_ - tprev_); tprev_ = t_; } } while (0)
#else
#define SPMM_T(i) do { } while (0)
#endif
template <bool BF, int CPT>
__global__ void __launch_bounds__(256)
spmm_list_kernel(const uint64_t* __restrict__ bits, const float* __restrict__ dinv, const float* __restrict__ vals,
                 const uint64_t* __restrict__ keep, const int32_t* __restrict__ goff, const float* __restrict__ x,
                 float* __restrict__ y, int R, int H, int slab, int cap, int transpose, int accumulate, int n, int nslab, int split, int seq,
                 const ZeroFill zf) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  // zero-fill ranges handed over by the caller (a few hundred KB: the first workgroups' threads cover them)
  if (zf.n0 > 0 || zf.n1 > 0) {
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4, st = (long long)gridDim.x * 256 * 4;
    for (long long i = i0; i < zf.n0; i += st) *reinterpret_cast<float4*>(zf.p0 + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long i = i0; i < zf.n1; i += st) *reinterpret_cast<float4*>(zf.p1 + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int W = (R + 63) / 64;
  float4* xs = reinterpret_cast<float4*>(dsm);                                                // [R][slab] fp32 rows ...
  const uint2* xs16 = reinterpret_cast<const uint2*>(dsm);                                    // ... or (BF) bf16 rows, 8 B per float4 column
  uint2* ent = reinterpret_cast<uint2*>(dsm + (size_t)R * slab * (BF ? 8 : 16));              // [cap] {j, w}
  unsigned long long* rb = reinterpret_cast<unsigned long long*>(ent);                        // [R][W] refined bit rows (fallback only; cap >= R*W)
  int* st = reinterpret_cast<int*>(ent + cap);                                                // [R + 1] row starts
  float* dv = reinterpret_cast<float*>(st + R + 1);                                           // [R]
  uint2* itm = reinterpret_cast<uint2*>(dv + R + 1);                                          // [R] work items ((2R + 1) floats behind ent: + 1 for 8-byte alignment)
  __shared__ int wsum[4];
  __shared__ int wsum2[4];                                                                    // groups of hub rows | of mid rows << 16
  __shared__ int ctr[2];                                                                      // extra items, scratch slots
  // Work decode (1-D grid).  A workgroup owns `seq` consecutive slabs of one graph and walks them one after the other: bit
  // rows, scan and edge list are paid once per workgroup (measured: ~8 us of set-up latency chain against ~5 us of data
  // movement per fp32 slab at h = 300).  Workgroups go round-robin over the 8 XCDs, each with its own L2: the chunks of
  // one graph sit next to each other in ONE XCD's queue, so the 128-byte lines that straddle a slab boundary and the
  // graph's bit rows / dinv / keep words are fetched from HBM once and hit that L2 for the sibling workgroups.
  int g, sl;
  {
    const int nchunk = (nslab + seq - 1) / seq;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    sl = (j % nchunk) * seq;
    g = (j / nchunk) * 8 + xcd;
    if (g >= n) return;
  }
#ifdef GH_MEASURE
  unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int c0 = sl * slab;
  // exact n / d for n, d < 2^16 by one multiply-high (a runtime integer division costs ~20 VALU instructions)
  auto magic_of = [](int d) { return (unsigned)(0xFFFFFFFFu / (unsigned)d) + 1u; };
  auto fdiv = [](int n, unsigned magic) { return (int)__umulhi((unsigned)n, magic); };
  const int H4 = H / 4;
  int ncol = min(slab, H4 - c0);
  const int row0 = goff ? goff[g] : g * R;
  const int NR = goff ? goff[g + 1] - row0 : R;
  if (NR <= 0) return;
  // ---- stage.  Issue order = completion order on the vector-memory counter: the few small loads of this thread's row
  // (bit words, keep words, dinv) go first, the slab's 16-byte loads after them, so the list is built from the former
  // while the latter are still in flight.  sched_barrier pins that order.
  const int trow = min(tid, R - 1);
  unsigned long long mrow[4] = {0ull, 0ull, 0ull, 0ull}, kwd[4] = {~0ull, ~0ull, ~0ull, ~0ull};
#pragma unroll
  for (int w = 0; w < 4; ++w)
    if (w < W) {
      mrow[w] = bits[((size_t)g * R + trow) * W + w];
      if (keep) kwd[w] = keep[(size_t)g * W + w];
    }
  const float dvv = vals ? 0.f : dinv[(size_t)g * R + trow];
  __builtin_amdgcn_sched_barrier(0);
  // LDS-DMA (buffer_load_dwordx4 ... lds): the LDS image is linear in (row, column), so a wave's 64 lanes land in 1 KB of
  // consecutive LDS; no staging registers, no ds_write.  fp32: one 16-byte chunk = one float4 column.  BF: the image
  // stays bf16 (half the LDS, twice the slab), one chunk = two columns (the launcher keeps slabs even).
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(x) + (size_t)row0 * H * (BF ? 2 : 4)), 0, 0x7fffffff, 0x00020000);
  auto dma_slab = [&]() __attribute__((always_inline)) {
    const int cpr = BF ? ncol / 2 : ncol;                 // chunks per row
    const int chunks = NR * cpr;
    const unsigned mg_cpr = magic_of(cpr);
    for (int base = 0; base < chunks; base += 256) {
      const int it = base + tid;
      if (it < chunks) {
        const int i = fdiv(it, mg_cpr), c = it - i * cpr;
        const int off = BF ? (i * H + (c0 + 2 * c) * 4) * 2 : (i * H + (c0 + c) * 4) * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xs + base + wave * 64), 16, off, 0, 0, 0);
      }
    }
  };
#ifdef GH_MEASURE
  const bool dbg_noagg = (split & 256) != 0, dbg_nodma = (split & 512) != 0;      // tool build: one half of the kernel at a time
  split &= 255;
  if (!dbg_nodma)
#endif
  dma_slab();
  __builtin_amdgcn_sched_barrier(0);
  SPMM_T(0);        // loads issued
  // this thread's row (tid < NR): refined words, degree
  int deg = 0;
  {
    bool ki = true;
    if (keep) {
      unsigned long long kw = kwd[0];
#pragma unroll
      for (int w = 1; w < 4; ++w) if ((trow >> 6) == w) kw = kwd[w];
      ki = (kw >> (trow & 63)) & 1ull;
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned long long m = (w < W && tid < NR) ? mrow[w] : 0ull;
      if (!ki) m &= kwd[w];                                    // edge survives iff keep(i) || keep(j)
      mrow[w] = m;
      deg += __popcll(m);
    }
  }
  if (tid < R) dv[tid] = dvv;
  // exclusive scan of deg over the 256 threads (and, alongside, the totals of the 8-edge groups of hub / mid-degree rows)
  constexpr int GS = 8;
  const int grp = (deg + GS - 1) / GS;
  const bool hub = deg > 4 * GS, mid = deg > 2 * GS && !hub;
  int inc = deg, inc2 = hub ? grp : (mid ? grp << 16 : 0);
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o), v2 = __shfl_up(inc2, o);
    if (lane >= o) { inc += v; inc2 += v2; }
  }
  if (lane == 63) { wsum[wave] = inc; wsum2[wave] = inc2; }
  if (tid < 2) ctr[tid] = 0;
  SPMM_T(1);        // row words arrived, scan
  __syncthreads();
  SPMM_T(2);        // barrier 1
  int base = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) if (w < wave) base += wsum[w];
  const int nnz = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  const int start = base + inc - deg;
  const bool listed = nnz <= cap;                 // workgroup-uniform
  if (tid <= R) st[tid] = (tid < NR) ? start : nnz;
  if (R >= 256 && tid == 0) st[R] = nnz;
  // Work items.  A word graph has a few hub nodes (frequent words: degree up to ~R/2 against a mean of ~6); with one thread per
  // (row, column group) the hub's threads run 8x longer than the rest and the workgroup waits for them.
  // NUMERICS (every row, split or not, every layout): a row's edges are summed in groups of GS = 8 -- each group from zero
  // in edge order, the group sums then added in group order.  SCHEDULING: a row with more than two groups may hand each
  // group to an item of its own; the group sums then travel through scratch slots in the part of the slab image that the
  // graph's R - NR absent rows leave unused and are added by a short second pass.  Which rows split (hubs first, then the
  // mid-degree rows, as far as the slots reach) never changes a result bit.
  const int spare = split ? (BF ? (R - NR) / 2 : (R - NR)) : 0;
  const int hm = wsum2[0] + wsum2[1] + wsum2[2] + wsum2[3];
  const bool split_hub = listed && (hm & 0xffff) > 0 && (hm & 0xffff) <= spare;
  const bool split_mid = listed && (hm >> 16) > 0 && (hm & 0xffff) + (hm >> 16) <= spare;
  if (tid < NR) {
    if (listed) {
      // item tid = the row itself (or its first group); a split row's further groups are appended behind the NR rows
      const bool sp = (hub && split_hub) || (mid && split_mid);
      const int pcs = sp ? grp : 1;
      int k0 = 0, s0 = 0;
      if (sp) { k0 = NR + atomicAdd(&ctr[0], pcs - 1) - 1; s0 = atomicAdd(&ctr[1], pcs); }
      for (int p = 0; p < pcs; ++p) {
        const int eb = start + p * GS, ee = sp ? min(start + deg, eb + GS) : start + deg;
        itm[p == 0 ? tid : k0 + p] = make_uint2((unsigned)tid | ((unsigned)p << 8) | ((unsigned)pcs << 14) | ((unsigned)s0 << 20),
                                                (unsigned)eb | ((unsigned)ee << 16));
      }
      const float di = dvv;
      int e = start;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (w < W) {
          unsigned long long m = mrow[w];
          while (m) {
            const int j = (w << 6) + __builtin_ctzll(m);
            m &= m - 1;
            const float wt = vals ? (transpose ? vals[((size_t)g * R + j) * R + tid] : vals[((size_t)g * R + tid) * R + j])
                                  : di * dv[j];
            ent[e++] = make_uint2((unsigned)j, __builtin_bit_cast(unsigned, wt));
          }
        }
    } else {
#pragma unroll
      for (int w = 0; w < 4; ++w) if (w < W) rb[tid * W + w] = mrow[w];
    }
  }
  SPMM_T(3);          // list build (thread 0's own row)
  const int sl_end = min(nslab, sl + seq);
  for (int s_ = sl; s_ < sl_end; ++s_) {
  if (s_ > sl) {                // next slab of this graph: every thread is done with the previous image and its scratch slots
    __syncthreads();
    c0 = s_ * slab;
    ncol = min(slab, H4 - c0);
#ifdef GH_MEASURE
    if (!dbg_nodma)
#endif
    dma_slab();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the slab has landed (LDS-DMA is tracked by vmcnt)
  SPMM_T(4);          // slab wait
  __syncthreads();
  SPMM_T(5);          // barrier 2 (the slowest list builder / DMA)
#ifdef GH_MEASURE
  if (dbg_noagg) continue;
#endif
  // ---- aggregate: thread = (item, column group q); columns q, q + TPR, ...
  const int TPR = (ncol + CPT - 1) / CPT;
  const unsigned mg_tpr = magic_of(TPR);
  const float* vg = vals ? vals + (size_t)g * R * R : nullptr;
  auto store_row = [&](int i, int q, const int* cc, const float4* acc) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      if (q + k * TPR >= ncol) break;
      float4 a = acc[k];
      if (BF) {
        uint2* o = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(y) + ((size_t)row0 + i) * H) + c0 + cc[k];
        if (accumulate) { const float4 p = bf4_to_f4(*o); a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w; }
        *o = f4_to_bf4(a);
      } else {
        float4* o = reinterpret_cast<float4*>(y + ((size_t)row0 + i) * H) + c0 + cc[k];
        if (accumulate) { const float4 p = *o; a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w; }
        *o = a;
      }
    }
  };
  if (listed) {
    float4* scr = reinterpret_cast<float4*>(dsm + (size_t)NR * ncol * (BF ? 8 : 16));      // [slot][ncol] fp32 partial rows
    const int nitems = NR + ctr[0];
    const bool any_split = ctr[1] > 0;            // workgroup-uniform
    const int nit = nitems * TPR;
    for (int it = tid; it < nit; it += 256) {
      const int kk = fdiv(it, mg_tpr), q = it - kk * TPR;
      const uint2 im = itm[kk];
      const int i = im.x & 255, pcs = (im.x >> 14) & 63;
      int cc[CPT];
      float4 acc[CPT];
#pragma unroll
      for (int k = 0; k < CPT; ++k) { cc[k] = min(q + k * TPR, ncol - 1); acc[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
      const int e1 = im.y >> 16, e00 = im.y & 0xffff;
      const int slot0 = (int)(im.x >> 20) + (int)((im.x >> 8) & 63);
      float4 part[CPT];
#pragma unroll
      for (int k = 0; k < CPT; ++k) part[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int e = e00; e < e1; e += 2) {
        const uint2 ea = ent[e];
        const bool two = e + 1 < e1;
        const uint2 eb = ent[two ? e + 1 : e];
        const float wa = __builtin_bit_cast(float, ea.y);
        const float wb = two ? __builtin_bit_cast(float, eb.y) : 0.f;
        float4 xa[CPT], xb[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
          if (BF) { xa[k] = bf4_to_f4(xs16[ea.x * ncol + cc[k]]); xb[k] = bf4_to_f4(xs16[eb.x * ncol + cc[k]]); }
          else { xa[k] = xs[ea.x * ncol + cc[k]]; xb[k] = xs[eb.x * ncol + cc[k]]; }
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
          part[k].x += wa * xa[k].x; part[k].y += wa * xa[k].y; part[k].z += wa * xa[k].z; part[k].w += wa * xa[k].w;
          part[k].x += wb * xb[k].x; part[k].y += wb * xb[k].y; part[k].z += wb * xb[k].z; part[k].w += wb * xb[k].w;
        }
        if ((((e - e00) & (GS - 2)) == GS - 2) || e + 2 >= e1) {        // the group's last pair: fold the group sum into the row sum
#pragma unroll
          for (int k = 0; k < CPT; ++k) {
            acc[k].x += part[k].x; acc[k].y += part[k].y; acc[k].z += part[k].z; acc[k].w += part[k].w;
            part[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
      if (pcs == 1) {
        store_row(i, q, cc, acc);
      } else {                     // a split row's item holds exactly one group: acc = 0 + part
#pragma unroll
        for (int k = 0; k < CPT; ++k)
          if (q + k * TPR < ncol) scr[slot0 * ncol + cc[k]] = acc[k];
      }
    }
    if (any_split) {
      __syncthreads();
      for (int it = tid; it < nit; it += 256) {
        const int kk = fdiv(it, mg_tpr), q = it - kk * TPR;
        const uint2 im = itm[kk];
        const int pcs = (im.x >> 14) & 63;
        if (pcs == 1 || ((im.x >> 8) & 63) != 0) continue;
        const int i = im.x & 255, s0 = (int)(im.x >> 20);
        int cc[CPT];
        float4 acc[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) { cc[k] = min(q + k * TPR, ncol - 1); acc[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
        for (int p = 0; p < pcs; ++p) {
#pragma unroll
          for (int k = 0; k < CPT; ++k) {
            const float4 v = scr[(s0 + p) * ncol + cc[k]];
            acc[k].x += v.x; acc[k].y += v.y; acc[k].z += v.z; acc[k].w += v.w;
          }
        }
        store_row(i, q, cc, acc);
      }
    }
  } else {
    const int nit = NR * TPR;
    for (int it = tid; it < nit; it += 256) {
      const int i = fdiv(it, mg_tpr), q = it - i * TPR;
      int cc[CPT];
      float4 acc[CPT];
#pragma unroll
      for (int k = 0; k < CPT; ++k) { cc[k] = min(q + k * TPR, ncol - 1); acc[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
      const float di = vals ? 0.f : dv[i];
      for (int w = 0; w < W; ++w) {
        unsigned long long m = rb[i * W + w];
        while (m) {
          const int j = (w << 6) + __builtin_ctzll(m);
          m &= m - 1;
          const float wt = vg ? (transpose ? vg[(size_t)j * R + i] : vg[(size_t)i * R + j]) : di * dv[j];
#pragma unroll
          for (int k = 0; k < CPT; ++k) {
            const float4 xv = BF ? bf4_to_f4(xs16[j * ncol + cc[k]]) : xs[j * ncol + cc[k]];
            acc[k].x += wt * xv.x; acc[k].y += wt * xv.y; acc[k].z += wt * xv.z; acc[k].w += wt * xv.w;
          }
        }
      }
      store_row(i, q, cc, acc);
    }
  }
  }     // slabs
  SPMM_T(6);          // aggregate + stores issued (thread 0)
}

// ---------------------------------------------------------------------------------------------------------------------
// Aggregation of the bf16 storage pipeline ON THE MATRIX PIPE (round 6).  The edge-list kernel above is VALU-issue-bound on bf16 rows
// (PMC on configs[4]: 44 M wave VALU instructions per launch = 72 us of a 122 us launch; every (edge, four columns) item costs a list
// read, an address, four unpack instructions and two packed FMAs), while the same sum as a DENSE product  y_g = A_g x_g  per graph
// (A_g: <= 128 x 128 normalised adjacency, 7 % dense; x_g: the graph's rows) is 2 R^2 h flops = 15 MFLOP per graph at h = 768 -- a
// few microseconds of v_mfma_f32_16x16x32_bf16 even though 93 % of the products are with zeros.
//   * A never exists in memory: lane (l15, q) BUILDS its MFMA operand -- row i = 16 mi + l15, columns 32 ks + 8 q .. + 7 -- from the
//     row's refined bit words and the graph's d^-1/2 values (or the dense values), once per workgroup, in registers.  The fp32
//     weights are split THREE ways into bf16 (hi + mid + lo: w - hi and r1 - mid are exact in fp32, the remainder after lo is below
//     2^-24 |w|), so the products with the bf16 activations are exact to fp32 precision and the accumulation is fp32: the result is
//     the edge-list kernel's up to fp32 summation order, then rounded ONCE to bf16.
//   * x_g is the B operand, k-major as it sits in memory: slabs of 128 columns (256 B per row) by LDS-DMA, fragments by two
//     ds_read_b64_tr_b16 each, 32-byte units XOR-swizzled on the source side exactly as in gemm_tn_pp.hip.h (256-byte pitch).  Rows
//     between NR and the next multiple of 32 are DMA'd with the out-of-range offset (zeros: 0 x stale LDS could be NaN).
//   * Wave w owns the row tiles mi = w and w + 4 (R <= 128) and walks the slab's column tiles; only ceil(NR / 32) k-steps and
//     ceil(NR / 16) row tiles are computed, so the work follows NR^2 in the node-compact layout.
// Preconditions (launch_spmm): bf16 rows, R <= 128, h % 8 == 0, 16-byte aligned rows.
// PIPE: slabs of 64 columns (128 B per row) in TWO 16 KB buffers -- the next slab's DMA is in flight under the current slab's MFMAs (same LDS
// footprint as one 128-column slab; 32-byte units swizzled with ((r >> 1) & 1) | ((r >> 3) & 1) << 1: rows alternate between the two bank halves)
template <int WAVES, bool PIPE>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 8 ? 1 : 2)
spmm_mfma_bf16_kernel(const uint64_t* __restrict__ bits, const float* __restrict__ dinv, const float* __restrict__ vals,
                      const uint64_t* __restrict__ keep, const int32_t* __restrict__ goff, const unsigned short* __restrict__ x,
                      unsigned short* __restrict__ y, int R, int H, int transpose, int accumulate, int n, int nslab, int seq) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  constexpr int NT = 8 / WAVES, NTHR = WAVES * 64;      // row tiles per wave: wave w owns mi = w + WAVES t
  constexpr int SC = PIPE ? 64 : 128, PITCH = SC * 2, CPR = PITCH / 16, BUFB = 128 * PITCH;      // slab columns, bytes per row, 16-byte chunks per row
  __shared__ __attribute__((aligned(16))) unsigned char xs[128 * 256];      // [buffer][row][PITCH], rows 0 .. KR - 1 of a slab
  __shared__ float dv[128];
  constexpr unsigned OOB = 0x80000000u;
  int g, sl;
  {
    const int nchunk = (nslab + seq - 1) / seq;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    sl = (j % nchunk) * seq;
    g = (j / nchunk) * 8 + xcd;
    if (g >= n) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  // row-tile owner: rotated by the graph's position in its XCD queue.  In the node-compact layout most graphs have five row tiles (65 nodes
  // on average), i.e. ONE wave with two tiles, and wave w of every workgroup runs on SIMD w.  (Measured: no effect -- the tile loop is not
  // bound by the matrix pipe of one SIMD; kept because it costs nothing.)
  const int wv = (wave + (g >> 3)) % WAVES;
  const int W = (R + 63) / 64;                 // 1 or 2 words per bit row
  const int row0 = goff ? goff[g] : g * R;
  const int NR = goff ? goff[g + 1] - row0 : R;
  if (NR <= 0) return;
  const int nks = (NR + 31) >> 5;              // k-steps of 32 graph rows
  const int KR = nks * 32;

  // ---- the slab DMA (issued first: the operand build below runs under its flight)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)row0 * H), 0, 0x7fffffff, 0x00020000);
  auto dma_slab = [&](int c0, int buf) __attribute__((always_inline)) {
    const int chunks = KR * CPR;
    for (int base = 0; base < chunks; base += NTHR) {
      const int it = base + tid;                       // (chunks is a multiple of the workgroup size: no partial iteration)
      const int i = it / CPR, sl16 = it % CPR;
      const int fz = PIPE ? ((((i >> 1) & 1) | (((i >> 3) & 1) << 1)) << 1) : (((i & 3) << 1) | (((i >> 3) & 1) << 3));
      const int c = sl16 ^ fz;
      const int col = c0 + 8 * c;
      const unsigned off = (i < NR && col < H) ? (unsigned)(i * H + col) * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xs + buf * BUFB + (base + wave * 64) * 16), 16, off, 0, 0, 0);
    }
  };
  // Issue order = completion order on the vector-memory counter: the few small loads (bit rows, keep words, d^-1/2) go FIRST, the slab's
  // DMA after them, so that the operand build below waits for the former with a counted vmcnt while the latter is still in flight.
  const float dvv = (tid < 128 && !vals && tid < NR) ? dinv[(size_t)g * R + tid] : 0.f;
  unsigned long long mw0[NT], mw1[NT], kw0 = ~0ull, kw1 = ~0ull;
  if (keep) { kw0 = keep[(size_t)g * W]; if (W > 1) kw1 = keep[(size_t)g * W + 1]; }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int i = 16 * (wv + WAVES * t) + l15;
    mw0[t] = 0ull; mw1[t] = 0ull;
    if (i < NR) {
      mw0[t] = bits[((size_t)g * R + i) * W];
      if (W > 1) mw1[t] = bits[((size_t)g * R + i) * W + 1];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  dma_slab(sl * SC, 0);
  __builtin_amdgcn_sched_barrier(0);

  // ---- d^-1/2 of the graph's nodes (pattern mode) for everybody
  if (tid < 128) dv[tid] = dvv;
  // (a bare barrier: __syncthreads() would also drain vmcnt, i.e. wait for the slab)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- A operands of this wave's row tiles, three bf16 pieces each: aF[t][ks][piece]
  uint4 aF[NT][4][3];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int mi = wv + WAVES * t;
    const int i = 16 * mi + l15;
    const bool live = i < NR;
    unsigned long long m0 = mw0[t], m1 = mw1[t];
    if (live && keep) {
      const bool ki = (((i < 64 ? kw0 : kw1) >> (i & 63)) & 1ull) != 0;
      if (!ki) { m0 &= kw0; m1 &= kw1; }                // edge survives iff keep(i) || keep(j)
    }
    const float di = live && !vals ? dv[i] : 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int j0 = 32 * ks + 8 * q;
      const unsigned byte = (unsigned)(((ks < 2 ? m0 : m1) >> ((32 * (ks & 1)) + 8 * q)) & 0xffull);
      float w[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool on = (byte >> b) & 1u;
        float v = 0.f;
        if (vals) { if (on) v = transpose ? vals[((size_t)g * R + (j0 + b)) * R + i] : vals[((size_t)g * R + i) * R + (j0 + b)]; }
        else v = on ? di * dv[j0 + b] : 0.f;
        w[b] = v;
      }
      unsigned hi[4], mid[4], lo[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float a0 = w[2 * b], a1 = w[2 * b + 1];
        const unsigned h2 = pack_bf2(a0, a1);
        const float r0 = a0 - __builtin_bit_cast(float, h2 << 16), r1 = a1 - __builtin_bit_cast(float, h2 & 0xffff0000u);
        const unsigned m2 = pack_bf2(r0, r1);
        const float s0 = r0 - __builtin_bit_cast(float, m2 << 16), s1 = r1 - __builtin_bit_cast(float, m2 & 0xffff0000u);
        hi[b] = h2; mid[b] = m2; lo[b] = pack_bf2(s0, s1);
      }
      aF[t][ks][0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      aF[t][ks][1] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
      aF[t][ks][2] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
  }
  const bool t0_live = 16 * wv < NR, t1_live = NT > 1 && 16 * (wv + WAVES) < NR;      // wave-uniform

  // ---- transpose-read base: lane (p = l15, g = q) points at row 8 q + (p >> 2) (+4: second read, +32 ks), four columns 4 (p & 3)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)xs;
  const int f5 = PIPE ? (((l15 >> 3) & 1) | ((q & 1) << 1)) : ((l15 >> 2) | ((q & 1) << 2));
  const unsigned rbase = lds0 + (unsigned)((8 * q + (l15 >> 2)) * PITCH + 8 * (l15 & 3));
  auto tr_read = [&](unsigned addr) __attribute__((always_inline)) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
  };

  const int sl_end = min(nslab, sl + seq);
  for (int s_ = sl; s_ < sl_end; ++s_) {
    const int c0 = s_ * SC;
    const unsigned bufo = PIPE ? (unsigned)(((s_ - sl) & 1) * BUFB) : 0u;
    if constexpr (PIPE) {
      // the other buffer was last read in the previous iteration, which ended with a barrier: restage it now.  The wait below lets
      // exactly the next slab's DMA instructions of this wave (KR * CPR / NTHR = nks of them) stay in flight; everything older -- this
      // slab's DMA and the previous slab's stores -- has then completed (vmcnt retires in order).
      if (s_ + 1 < sl_end) {
        dma_slab(c0 + SC, ((s_ - sl) & 1) ^ 1);
        static_assert(!PIPE || WAVES == 4, "the counted wait below assumes nks DMA instructions per wave and slab");
        if (nks == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (nks == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (nks == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    } else {
      if (s_ > sl) {
        __syncthreads();                    // every wave is done with the previous slab's image
        dma_slab(c0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const int ntile = (min(SC, H - c0) + 15) >> 4;
    // column tiles in pairs over two fragment sets: the next tile's transpose reads (and, accumulating, its y values) are in flight
    // while the current tile's MFMAs run
    auto load_b = [&](int ni, uint2* b0, uint2* b1) __attribute__((always_inline)) {
      const unsigned ad = rbase + bufo + (unsigned)(((ni ^ f5) & (PIPE ? 3 : 7)) << 5);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (ks < nks) { b0[ks] = tr_read(ad + ks * 32 * PITCH); b1[ks] = tr_read(ad + ks * 32 * PITCH + 4 * PITCH); }
        else { b0[ks] = make_uint2(0u, 0u); b1[ks] = make_uint2(0u, 0u); }
    };
    auto tile = [&](int ni, uint2* b0, uint2* b1, uint2* nb0, uint2* nb1) __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]) :: "memory");
      if (ni + 1 < ntile) load_b(ni + 1, nb0, nb1);
      // acc[r] = y[row 16 mi + l15][column c0 + 16 ni + 4 q + r]: four consecutive bf16 = one 8-byte access
      const int col = c0 + 16 * ni + 4 * q;
      const int i0r = 16 * wv + l15, i1r = i0r + 16 * WAVES;
      const bool ok0 = i0r < NR && col < H, ok1 = NT > 1 && i1r < NR && col < H;
      uint2* o0 = reinterpret_cast<uint2*>(y + ((size_t)row0 + i0r) * H + col);
      uint2* o1 = reinterpret_cast<uint2*>(y + ((size_t)row0 + i1r) * H + col);
      uint2 p0 = make_uint2(0u, 0u), p1 = make_uint2(0u, 0u);
      if (accumulate & 1) { if (ok0) p0 = *o0; if (ok1) p1 = *o1; }
      f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < nks) {
          const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, make_uint4(b0[ks].x, b0[ks].y, b1[ks].x, b1[ks].y));
          // (smallest pieces first: the fp32 accumulator sees lo + mid before hi)
          if (t0_live) {
#pragma unroll
            for (int pc = 2; pc >= 0; --pc) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, __builtin_bit_cast(bf16x8_t, aF[0][ks][pc]), acc0, 0, 0, 0);
          }
          if constexpr (NT > 1) {
            if (t1_live) {
#pragma unroll
              for (int pc = 2; pc >= 0; --pc) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, __builtin_bit_cast(bf16x8_t, aF[NT - 1][ks][pc]), acc1, 0, 0, 0);
            }
          }
        }
      }
#ifdef GH_MEASURE
      if (accumulate & 256) {      // tool build: everything but the stores
        if (acc0[0] + acc1[0] == 12345.678f) *o0 = make_uint2(0u, 0u);
        return;
      }
#endif
      if (ok0) {
        float4 v = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
        if (accumulate & 1) { const float4 pv = bf4_to_f4(p0); v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w; }
        *o0 = f4_to_bf4(v);
      }
      if (ok1) {
        float4 v = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
        if (accumulate & 1) { const float4 pv = bf4_to_f4(p1); v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w; }
        *o1 = f4_to_bf4(v);
      }
    };
#ifdef GH_MEASURE
    if (accumulate & 512) { if constexpr (PIPE) asm volatile("s_barrier" ::: "memory"); continue; }      // tool build: set-up + slab DMA only
#endif
    uint2 bA0[4], bA1[4], bB0[4], bB1[4];
    load_b(0, bA0, bA1);
    for (int ni = 0; ni < ntile; ni += 2) {
      tile(ni, bA0, bA1, bB0, bB1);
      if (ni + 1 < ntile) tile(ni + 1, bB0, bB1, bA0, bA1);
    }
    if constexpr (PIPE) asm volatile("s_barrier" ::: "memory");      // (every transpose read of this buffer was waited for in tile())
  }
#endif
}

#ifdef GH_MEASURE
extern "C" int gh_debug_spmm_phases(unsigned* out, int reset) {     // out: [8192][8]
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spmm_phase), 8192 * 8 * sizeof(unsigned)) != hipSuccess) return 1;
  if (reset) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_spmm_phase)) != hipSuccess || hipMemset(p, 0, 8192 * 8 * sizeof(unsigned)) != hipSuccess) return 1; }
  return 0;
}
#endif

int launch_spmm(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep, const int32_t* goff,
                int m_real, const float* x, float* y, int n, int r, int h, int transpose, int accumulate, hipStream_t s, int bf16,
                const ZeroFill* zfp, bool* zf_done) {
  if (zf_done) *zf_done = false;
  GH_REQUIRE(r <= MAX_R, "spmm: padded graph size %d > %d", r, MAX_R);
  GH_REQUIRE(vals || dinv, "spmm: need dinv or vals");
  const int W = words_for(r);
  const bool v4 = (h % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  GH_REQUIRE(!bf16 || v4, "spmm: the bf16 variant needs h %% 4 == 0 and 16-byte aligned rows");
  const int V = v4 ? 4 : 1;
  const int hv = h / V;
  // slab: <= 32 float4 (or 128 scalars) per row, as even as possible
  // (sized so that a workgroup's slab stays under ~32 KB of LDS: four to five workgroups per CU, also at R = 200)
  static int cap_env = -2;
  if (cap_env == -2) cap_env = measure_env("GH_SPMM_SLAB_KB", -1);
  // round 4, A/B on one box: many graphs of <= 128 nodes (the bench step's 960 x 100) run 0.4 % faster per STEP on 24 KB slabs (five
  // slabs of 15 float4 instead of four of 19), few graphs (218: the realistic step) 1 % slower
  const int cap_kb = cap_env > 0 ? cap_env : ((goff && n >= 512 && r <= 128 && !bf16) ? 24 : 32);      // (node-compact layout only: the padded 100-row graphs run 58 -> 69 us on 24 KB slabs)   // measured per step: 48 KB -> 0.314 ms, 32 -> 0.288, 24 -> 0.294, 16 -> 0.344 (R = 100); R = 200: 0.60 -> 0.43
  const int lds_cap = (cap_kb * 1024) / (r * (v4 ? 16 : 4));
  const int slab_max = v4 ? (lds_cap < 32 ? (lds_cap < 4 ? 4 : lds_cap) : 32) : (lds_cap < 128 ? (lds_cap < 16 ? 16 : lds_cap) : 128);
  const int nslab = (hv + slab_max - 1) / slab_max;
  const int slab = (hv + nslab - 1) / nslab;
  const size_t lds = ((((size_t)r * slab * V) + 3) & ~(size_t)3) * 4 + (size_t)r * W * 8 + (size_t)r * 4;
  dim3 grid(n, nslab);
  // algorithmic bytes: x in + y out (+ y in when accumulating) + bit rows + dinv (or the touched dense values)
  const double rows = goff ? (double)m_real : (double)n * r;
  const double alg_bytes = (2.0 + (accumulate ? 1.0 : 0.0)) * rows * h * (bf16 ? 2.0 : 4.0) +
                           (double)n * ((double)r * W * 8.0 + (vals ? (double)r * r * 4.0 : (double)r * 4.0));
  // 4 (default; 3 / 5: one / three columns per thread): edge-list slab kernel, LDS-DMA staging -- 62 us vs 69 us for the
  // bit-walk slab kernel (0) on 960 x 100 x 300 Zipf word graphs, 50 vs 61 us on hub-free graphs;
  // (a graph-per-workgroup pipeline over two LDS buffers, next slab's DMA in flight during the aggregation, measured
  // slower -- 74-85 us at two workgroups per CU -- and was dropped);
  // 1 / 2: LDS-free gather variants (thread-per-float4 / wave-per-row).  Measured equal or slower on MI355X: with
  // ~8K waves in flight their sliding-window working set (~49 MB) thrashes the 32 MB of L2 (hit rate 32 %).
  static int variant = -1;
  if (variant < 0) { variant = measure_env("GH_SPMM_VARIANT", 4); if (variant > 5) variant = 4; }
  // bf16 rows: columns per thread of the list kernel (tool build: GH_SPMM_BF16_CPT).  A bf16 column is an 8-byte LDS read and four
  // unpack instructions in front of its FMAs, the per-edge overhead (list entry, address) weighs more than in fp32:
  // 960 graphs x h = 768, window 5: 148.6 / 126.9 / 120.0 / 122.2 / 126.2 us at 1 / 2 / 3 / 4 / 6 columns per thread (fewer, longer work
  // items per row beyond three: the last round of items is thinly filled)
  static int bf_cpt = -1;
  if (bf_cpt < 0) { bf_cpt = measure_env("GH_SPMM_BF16_CPT", 3); if (bf_cpt < 1 || bf_cpt > 3) bf_cpt = 3; }
  const int ptag = n < PROF_FEW_GROUPS ? PROF_FEW_ROWS : PROF_SPMM;
  prof_begin(s, ptag);
  static int mfma_agg = -1;
  if (mfma_agg < 0) mfma_agg = measure_env("GH_SPMM_MFMA", 1);
  if (bf16 && mfma_agg && r <= 128 && h % 8 == 0 && variant >= 3) {
    // bf16 rows: the aggregation as a dense product per graph on the matrix pipe (spmm_mfma_bf16_kernel)
    static int mseq = -1, mw = -1, mpipe = -1;
    if (mseq < 0) mseq = measure_env("GH_SPMM_MFMA_SEQ", 3);
    if (mw < 0) mw = measure_env("GH_SPMM_MFMA_WAVES", 4);
    if (mpipe < 0) mpipe = measure_env("GH_SPMM_MFMA_PIPE", 0);
    const bool pipe = mpipe != 0 && mw != 8;
    const int sc = pipe ? 64 : 128;
    const int ns = (h + sc - 1) / sc;
    int seq = n < 256 ? 1 : (pipe ? 2 * mseq : mseq);
    if (seq > ns) seq = ns;
    if (seq < 1) seq = 1;
    const int nchunk = (ns + seq - 1) / seq;
    const unsigned short* x16 = reinterpret_cast<const unsigned short*>(x);
    unsigned short* y16 = reinterpret_cast<unsigned short*>(y);
    const dim3 mgrid(((n + 7) / 8) * 8 * nchunk);
#ifdef GH_MEASURE
    static int mdbg = -1;
    if (mdbg < 0) mdbg = measure_env("GH_SPMM_MFMA_NOSTORE", 0);
    if (mdbg) accumulate |= 256 * mdbg;
#endif
    // (measured equal and kept in the tool build only, DESIGN 4.5: eight waves per workgroup with one row tile each -- GH_SPMM_MFMA_WAVES=8 --
    //  and 64-column slabs in two buffers with the next slab's DMA under the current slab's MFMAs -- GH_SPMM_MFMA_PIPE=1)
#ifdef GH_MEASURE
    if (mw == 8) hipLaunchKernelGGL((spmm_mfma_bf16_kernel<8, false>), mgrid, dim3(512), 0, s, bits, dinv, vals, keep, goff, x16, y16, r, h, transpose, accumulate, n, ns, seq);
    else if (pipe) hipLaunchKernelGGL((spmm_mfma_bf16_kernel<4, true>), mgrid, dim3(256), 0, s, bits, dinv, vals, keep, goff, x16, y16, r, h, transpose, accumulate, n, ns, seq);
    else
#endif
    hipLaunchKernelGGL((spmm_mfma_bf16_kernel<4, false>), mgrid, dim3(256), 0, s, bits, dinv, vals, keep, goff, x16, y16, r, h, transpose, accumulate, n, ns, seq);
  } else
  if (v4 && variant >= 3 && r <= 256 && (!bf16 || hv % 2 == 0)) {
    // edge-list kernel; variant 3: one column per thread, 4 (default): two, 5: three.  LDS pitch = slab columns.
    const int cap = 10 * r;                        // edges per graph the list holds (a window-5 word graph has <= 9 R)
    int lslab = slab;
    dim3 lgrid = grid;
    if (bf16) {                                    // bf16 LDS image: 8 B per column -> twice the columns per slab, kept even
      int smax = (cap_kb * 1024) / (r * 8);
      smax = smax > 64 ? 64 : (smax < 4 ? 4 : smax);
      smax &= ~1;
      const int ns = (hv + smax - 1) / smax;
      lslab = (((hv + ns - 1) / ns) + 1) & ~1;
      lgrid = dim3(n, (hv + lslab - 1) / lslab);
    }
    // Rows that are whole 128-byte lines (h = 768: 1536 B in bf16, 3072 B in fp32): slabs of whole lines.  The even split above
    // gives 40 bf16 columns = 320 B per row and slab -- 2.5 lines, every slab boundary inside a line that two workgroups fetch
    // (configs[4] bf16, A/B on one box: five slabs of 40 columns 0.675 ms of aggregation per step, four slabs of 48 columns =
    // 3 lines 0.607 ms; 64 columns = 51 KB of LDS per workgroup 0.91 ms).  Up to 40 KB of slab image per workgroup.
    {
      const int col_b = bf16 ? 8 : 16, per_line = 128 / col_b;            // float4 columns per 128-byte line
      static int aligned = -1;
      if (aligned < 0) aligned = measure_env("GH_SPMM_LINE_SLABS", 1);
      if (aligned && cap_env <= 0 && ((size_t)h * (bf16 ? 2 : 4)) % 128 == 0 && hv % per_line == 0) {
        int best = 0;
        static int line_kb = -1;
        if (line_kb < 0) line_kb = measure_env("GH_SPMM_LINE_KB", 40);
        for (int c = per_line; c <= hv && (size_t)r * c * col_b <= (size_t)line_kb * 1024; c += per_line) best = c;
        if (best >= 2 * per_line || (line_kb != 40 && best >= per_line)) {
          const int ns = (hv + best - 1) / best;
          int even = (((hv + ns - 1) / ns) + per_line - 1) / per_line * per_line;      // as even as whole lines allow
          lslab = even;
          lgrid = dim3(n, (hv + lslab - 1) / lslab);
        }
      }
    }
    const size_t llds = (size_t)r * lslab * (bf16 ? 8 : 16) + (size_t)cap * 8 + (size_t)(r + 1) * 4 + (size_t)r * 4 + 4 + (size_t)r * 8;   // + item table
    const void* fn;
    const int lv = variant > 5 ? 5 : variant;
    if (bf16) {
      const int c = lv == 3 ? 1 : lv == 5 ? 3 : bf_cpt;
      fn = c == 1 ? (const void*)spmm_list_kernel<true, 1> : c == 3 ? (const void*)spmm_list_kernel<true, 3> : (const void*)spmm_list_kernel<true, 2>;
    }
    else fn = lv == 3 ? (const void*)spmm_list_kernel<false, 1> : lv == 5 ? (const void*)spmm_list_kernel<false, 3> : (const void*)spmm_list_kernel<false, 2>;
    static bool attrl[6] = {false, false, false, false, false, false};
    const int ai = bf16 ? (3 + (lv == 3 ? 0 : lv == 5 ? 2 : bf_cpt - 1)) : (lv - 3);
    if (!attrl[ai] && llds > 64 * 1024) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attrl[ai] = true; }
    // slabs per workgroup, measured inside the bench step (ms of aggregation per step at 1 / 2 / all slabs per workgroup):
    //   960 graphs, R = 100, h = 300 fp32 (4 slabs): 0.302 / 0.282 / 0.262;   640 graphs, R = 200 (8 slabs): 0.536 / 0.467 / 0.426;
    //   960 graphs, h = 768 bf16 image, window 5 (5 slabs): 0.634 / 0.620 / 0.698 -- long per-slab work: the whole graph in one
    //   workgroup leaves one thinly balanced round.  Few-graph launches (claim side) keep one slab per workgroup: latency.
    static int split = -1, spw_env = -1;
    if (split < 0) split = measure_env("GH_SPMM_SPLIT", 1) | (measure_env("GH_SPMM_DBG", 0) << 8);
    if (spw_env < 0) spw_env = measure_env("GH_SPMM_SPW", 0);
    int ns_arg = (int)lgrid.y;
    int seq = spw_env > 0 ? spw_env : (n < 256 ? 1 : (bf16 ? 2 : ns_arg));
    if (seq > ns_arg) seq = ns_arg;
    lgrid = dim3(((n + 7) / 8) * 8 * ((ns_arg + seq - 1) / seq), 1);
    ZeroFill zf = {nullptr, 0, nullptr, 0};
    if (zfp && (zfp->n0 % 4 == 0) && (zfp->n1 % 4 == 0) && ((reinterpret_cast<uintptr_t>(zfp->p0) | reinterpret_cast<uintptr_t>(zfp->p1)) & 15) == 0) {
      zf = *zfp;
      if (zf_done) *zf_done = true;
    }
    void* args[] = {(void*)&bits, (void*)&dinv, (void*)&vals, (void*)&keep, (void*)&goff, (void*)&x, (void*)&y, (void*)&r, (void*)&h,
                    (void*)&lslab, (void*)&cap, (void*)&transpose, (void*)&accumulate, (void*)&n, (void*)&ns_arg, (void*)&split, (void*)&seq,
                    (void*)&zf};
    (void)hipLaunchKernel(fn, lgrid, dim3(256), args, llds, s);
  } else if (bf16) {
    static bool attrb = false;
    if (!attrb && lds > 64 * 1024) { (void)hipFuncSetAttribute((const void*)spmm_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attrb = true; }
    hipLaunchKernelGGL((spmm_kernel<4, true>), grid, dim3(256), lds, s, bits, dinv, vals, keep, goff, x, y, r, h, slab, transpose, accumulate);
  } else if (v4 && variant == 2 && h / 4 <= 128 && !goff) {
    constexpr int RPW = 5;
    const int wpg = (r + RPW - 1) / RPW;
    hipLaunchKernelGGL(spmm_wave_kernel<RPW>, dim3((n * wpg + 3) / 4), dim3(256), 0, s, bits, dinv, vals, keep, x, y, r, h,
                       transpose, accumulate, wpg);
  } else if (v4 && variant == 1 && !goff) {
    const int bpg = (r * (h / 4) + 2047) / 2048;      // ~8 float4 outputs per thread
    hipLaunchKernelGGL(spmm_gather_kernel, dim3(n * bpg), dim3(256), 0, s, bits, dinv, vals, keep, x, y, r, h, transpose,
                       accumulate, bpg);
  } else if (v4) {
    static bool attr4 = false;     // only raise the dynamic-LDS cap when a launch actually needs more than 64 KB
    if (!attr4 && lds > 64 * 1024) { (void)hipFuncSetAttribute((const void*)spmm_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr4 = true; }
    hipLaunchKernelGGL(spmm_kernel<4>, grid, dim3(256), lds, s, bits, dinv, vals, keep, goff, x, y, r, h, slab, transpose, accumulate);
  } else {
    static bool attr1 = false;
    if (!attr1 && lds > 64 * 1024) { (void)hipFuncSetAttribute((const void*)spmm_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr1 = true; }
    hipLaunchKernelGGL(spmm_kernel<1>, grid, dim3(256), lds, s, bits, dinv, vals, keep, goff, x, y, r, h, slab, transpose, accumulate);
  }
  prof_end(ptag, alg_bytes, s);
  GH_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// node-compact layout plan.  goff = exclusive prefix sum of the node counts (one workgroup, n <= a few
// thousand graphs), then one workgroup per graph fills the row maps.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
ragged_scan_kernel(const int32_t* __restrict__ n_nodes, int n, int R, int32_t* __restrict__ goff) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = min(tid * per, n), hi = min(lo + per, n);
  int sum = 0;
  for (int g = lo; g < hi; ++g) sum += min(max(n_nodes[g], 0), R);
  part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {          // Hillis-Steele inclusive scan of the per-thread sums
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - sum;
  for (int g = lo; g < hi; ++g) { goff[g] = run; run += min(max(n_nodes[g], 0), R); }
  if (tid == 1023) goff[n] = part[1023];
}

__global__ void __launch_bounds__(256)
ragged_fill_kernel(const int32_t* __restrict__ goff, const int32_t* __restrict__ node_ids, int n, int R,
                   int32_t* __restrict__ rowg, int32_t* __restrict__ src, int32_t* __restrict__ cids,
                   float* __restrict__ maskf) {
  const int g = blockIdx.x;
  const int row0 = goff[g], NR = goff[g + 1] - row0;
  const int pad0 = goff[n] + g * R - row0 - NR;
  for (int j = threadIdx.x; j < R; j += 256) {
    const int row = j < NR ? row0 + j : pad0 + j;
    rowg[row] = g;
    src[row] = g * R + j;
    const int id = (cids || maskf) ? node_ids[(size_t)g * R + j] : 0;
    if (cids) cids[row] = id;
    if (maskf) maskf[row] = id >= 1 ? 1.f : 0.f;       // the word attention's mask (doc >= 1, graph_based_semantic_structure.py:180)
  }
}

// The plan in ONE launch for n <= 4096 graphs: every workgroup sums the node counts in front of its graph itself (n loads of
// 4 bytes out of L2, 960 at the bench shape) instead of waiting for a one-workgroup scan kernel; optionally it also scatters
// its graph's node ids into the padded `document` tensor (gh_get_prepare).
__global__ void __launch_bounds__(256)
ragged_plan1_kernel(const int32_t* __restrict__ n_nodes, const int32_t* __restrict__ node_ids, int n, int R,
                    int32_t* __restrict__ goff, int32_t* __restrict__ rowg, int32_t* __restrict__ src, int32_t* __restrict__ cids,
                    float* __restrict__ maskf, const int64_t* __restrict__ slot, int32_t* __restrict__ document) {
  __shared__ int red[2][4];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int before = 0, all = 0;
  for (int i = tid; i < n; i += 256) {
    const int v = min(max(n_nodes[i], 0), R);
    all += v;
    if (i < g) before += v;
  }
  for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); all += __shfl_xor(all, o); }
  if (lane == 0) { red[0][wave] = before; red[1][wave] = all; }
  __syncthreads();
  const int row0 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  const int total = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const int NR = min(max(n_nodes[g], 0), R);
  if (tid == 0) { goff[g] = row0; if (g == n - 1) goff[n] = total; }
  const int pad0 = total + g * R - row0 - NR;
  const size_t dst = (slot && document) ? (size_t)slot[g] * R : 0;
  for (int j = tid; j < R; j += 256) {
    const int row = j < NR ? row0 + j : pad0 + j;
    rowg[row] = g;
    src[row] = g * R + j;
    const int id = (cids || maskf || document) ? node_ids[(size_t)g * R + j] : 0;
    if (cids) cids[row] = id;
    if (maskf) maskf[row] = id >= 1 ? 1.f : 0.f;
    if (slot && document) document[dst + j] = id;
  }
}

// ------------------------------------------------------------------------------------------------
// top-k keep set from R scores held in LDS: rank by counting, ballot packs the 64-node words.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void topk_keep(const float* ss, int R, int k, uint64_t* keep_out, int tid) {
  // one thread per node (R <= 256 = blockDim); wave w produces word w
  bool kept = false;
  if (tid < R) {
    const float si = ss[tid];
    int rank = 0;
    for (int j = 0; j < R; ++j) {
      const float sj = ss[j];
      rank += (sj > si) || (sj == si && j < tid);
    }
    kept = rank < k;
  }
  const unsigned long long m = __ballot(kept);
  const int W = (R + 63) / 64;
  if ((tid & 63) == 0 && (tid >> 6) < W) keep_out[tid >> 6] = m;
}

// word scorer (GGNN 300->1, wrapper.py:167) + GSL top-k (:216-219), one workgroup per graph
__global__ void __launch_bounds__(256)
scorer_gsl_kernel(const uint64_t* __restrict__ bits, const float* __restrict__ dinv, const float* __restrict__ vals,
                  const int32_t* __restrict__ goff, const float* __restrict__ feat, const float* __restrict__ xs_in,
                  const float* __restrict__ w_p, const float* __restrict__ gate, int R, int H, int k, int pads_collapsed, float* __restrict__ score, uint64_t* __restrict__ keep, unsigned drop_thresh,
                  float drop_scale, unsigned drop_seed, int xs_parts, long long xs_stride) {
  __shared__ float xs[MAX_R];
  __shared__ float ss[MAX_R];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = (R + 63) / 64;
  // node-compact layout: real node j < NR sits in feature row goff[g] + j; padding node j >= NR of graph g in row
  // goff[n] + (g*R - goff[g]) + (j - NR)  (all padding rows follow the real rows of the whole batch).  The padding
  // nodes still compete in the top-k with their own (bias- and dropout-driven) scores, as in wrapper.py:216-219.
  const int row0 = goff ? goff[g] : g * R;
  const int NR = goff ? goff[g + 1] - row0 : R;
  // pads_collapsed (evaluation mode: no dropout, so every padding row of the batch is the same vector): all padding
  // nodes read the single representative row goff[n]
  const int pad0 = goff ? goff[gridDim.x] + (pads_collapsed ? 0 : g * R - row0 - NR) : row0;
  // x_j = feat_j . w_p   (proj, no bias)
  const bool v4 = (H % 4 == 0) && ((reinterpret_cast<uintptr_t>(feat) & 15) == 0) && ((reinterpret_cast<uintptr_t>(w_p) & 15) == 0);
  if (xs_in) {      // projection already done by the producing cell's epilogue (gh_ggnn_cell_fwd score_x): one float per node
    if (tid < R) {      // (xs_parts > 1: one partial per column block of a wide cell output, added in block order)
      const unsigned xr = (unsigned)(tid < NR ? row0 + tid : (pads_collapsed ? pad0 : pad0 + tid));
      float v = xs_in[xr];
      for (int pp = 1; pp < xs_parts; ++pp) v += xs_in[(size_t)pp * xs_stride + xr];
      xs[tid] = v;
    }
  } else {
  for (int j = wave; j < R; j += 4) {
    float acc = 0.f;
    const unsigned frow = (unsigned)(j < NR ? row0 + j : (pads_collapsed ? pad0 : pad0 + j));
    const float* fg = feat + (size_t)frow * H;
    if (v4) {
      const float4* fr = reinterpret_cast<const float4*>(fg);
      const float4* wr = reinterpret_cast<const float4*>(w_p);
      for (int c = lane; c < H / 4; c += 64) {
        float4 a = fr[c];
        const float4 b = wr[c];
        if (drop_thresh) a = drop4(a, drop_seed, frow * (unsigned)H + 4u * c, drop_thresh, drop_scale);
        acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
      }
    } else {
      for (int c = lane; c < H; c += 64) {
        float a = fg[c];
        if (drop_thresh) a = drop_hash(drop_seed, frow * (unsigned)H + c) >= drop_thresh ? a * drop_scale : 0.f;
        acc += a * w_p[c];
      }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) xs[j] = acc;
  }
  }
  __syncthreads();
  if (tid < R) {
    const int i = tid;
    float a = 0.f;
    const float di = vals ? 0.f : dinv[(size_t)g * R + i];
    for (int w = 0; w < W; ++w) {
      unsigned long long m = bits[((size_t)g * R + i) * W + w];
      while (m) {
        const int j = (w << 6) + __builtin_ctzll(m);
        m &= m - 1;
        const float wt = vals ? vals[((size_t)g * R + i) * R + j] : di * dinv[(size_t)g * R + j];
        a += wt * xs[j];
      }
    }
    const float x = xs[i];
    const float z = 1.f / (1.f + expf(-((gate[0] * a + gate[1]) + (gate[2] * x + gate[3]))));
    const float r = 1.f / (1.f + expf(-((gate[4] * a + gate[5]) + (gate[6] * x + gate[7]))));
    const float hh = tanhf((gate[8] * a + gate[9]) + (gate[10] * (r * x) + gate[11]));
    const float sc = hh * z + x * (1.f - z);
    ss[i] = sc;
    score[(size_t)g * R + i] = sc;
  }
  __syncthreads();
  topk_keep(ss, R, k, keep + (size_t)g * W, tid);
}

__global__ void __launch_bounds__(256)
gsl_topk_kernel(const float* __restrict__ score, int R, int k, uint64_t* __restrict__ keep) {
  __shared__ float ss[MAX_R];
  const int g = blockIdx.x, tid = threadIdx.x;
  if (tid < R) ss[tid] = score[(size_t)g * R + tid];
  __syncthreads();
  topk_keep(ss, R, k, keep + (size_t)g * ((R + 63) / 64), tid);
}

}  // namespace gh

using namespace gh;

extern "C" int gh_graph_build(const int32_t* tokens, const int32_t* lengths, int n_texts, int fixed_length,
                              int window, int32_t* node_ids, int32_t* n_nodes, uint64_t* bits, float* dinv,
                              gh_stream_t stream) {
  GH_REQUIRE(fixed_length > 0 && fixed_length <= MAX_R, "graph_build: fixed_length %d not in [1,%d]", fixed_length, MAX_R);
  GH_REQUIRE(window >= 1, "graph_build: window %d < 1", window);
  if (n_texts <= 0) return 0;
  prof_begin((hipStream_t)stream, PROF_GRAPH_BUILD);
  hipLaunchKernelGGL(graph_build_kernel, dim3(n_texts), dim3(256), 0, (hipStream_t)stream, tokens, lengths,
                     fixed_length, window, node_ids, n_nodes, bits, dinv);
  prof_end(PROF_GRAPH_BUILD, (double)n_texts * (12.0 * fixed_length + 8.0 * fixed_length * words_for(fixed_length) + 8.0),
           (hipStream_t)stream);
  GH_LAUNCH_CHECK();
  return 0;
}

extern "C" int gh_ragged_plan(const int32_t* n_nodes, const int32_t* node_ids, int n, int r, int32_t* goff,
                              int32_t* rowg, int32_t* src, int32_t* cids, float* maskf, gh_stream_t stream) {
  GH_REQUIRE(r > 0 && r <= MAX_R, "ragged_plan: r=%d not in [1,%d]", r, MAX_R);
  GH_REQUIRE((cids == nullptr && maskf == nullptr) || (node_ids != nullptr), "ragged_plan: cids / maskf need node_ids");
  if (n <= 0) return 0;
  return launch_ragged_plan(n_nodes, node_ids, n, r, goff, rowg, src, cids, maskf, nullptr, nullptr, (hipStream_t)stream, nullptr);
}

// slot / document: also scatter the node ids of graph g into document[slot[g]][:] (the composite batch preparation)
int gh::launch_ragged_plan(const int32_t* n_nodes, const int32_t* node_ids, int n, int r, int32_t* goff, int32_t* rowg, int32_t* src,
                           int32_t* cids, float* maskf, const int64_t* slot, int32_t* document, hipStream_t s, bool* scattered) {
  if (scattered) *scattered = false;
  if (n <= 4096 && (node_ids || !(slot && document))) {
    hipLaunchKernelGGL(ragged_plan1_kernel, dim3(n), dim3(256), 0, s, n_nodes, node_ids, n, r, goff, rowg, src, cids, maskf, slot, document);
    GH_LAUNCH_CHECK();
    if (scattered) *scattered = slot && document;
    return 0;
  }
  hipLaunchKernelGGL(ragged_scan_kernel, dim3(1), dim3(1024), 0, s, n_nodes, n, r, goff);
  hipLaunchKernelGGL(ragged_fill_kernel, dim3(n), dim3(256), 0, s, goff, node_ids, n, r, rowg, src, cids, maskf);
  GH_LAUNCH_CHECK();
  return 0;      // (*scattered stays false: the document scatter, if any, is still the caller's)
}

int gh::launch_graph_build2(const int32_t* ta, const int32_t* la, int na, int ra, int32_t* ida, int32_t* nna, uint64_t* ba, float* da,
                            const int32_t* tb, const int32_t* lb, int nb, int rb, int32_t* idb, int32_t* nnb, uint64_t* bb, float* db,
                            int window, hipStream_t s) {
  GH_REQUIRE(ra > 0 && ra <= MAX_R && rb > 0 && rb <= MAX_R, "graph_build: fixed lengths %d / %d not in [1,%d]", ra, rb, MAX_R);
  GH_REQUIRE(window >= 1, "graph_build: window %d < 1", window);
  if (na + nb <= 0) return 0;
  const GraphSide A = {ta, la, ra, ida, nna, ba, da}, B = {tb, lb, rb, idb, nnb, bb, db};
  prof_begin(s, PROF_GRAPH_BUILD);
  hipLaunchKernelGGL(graph_build2_kernel, dim3(na + nb), dim3(256), 0, s, A, B, na, window);
  prof_end(PROF_GRAPH_BUILD, (double)na * (12.0 * ra + 8.0 * ra * words_for(ra) + 8.0) + (double)nb * (12.0 * rb + 8.0 * rb * words_for(rb) + 8.0), s);
  GH_LAUNCH_CHECK();
  return 0;
}

extern "C" int gh_adj_pack_f64(const double* adj, int n, int r, uint64_t* bits, float* vals, gh_stream_t stream) {
  GH_REQUIRE(r > 0 && r <= MAX_R, "adj_pack: r=%d not in [1,%d]", r, MAX_R);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(adj_pack_kernel<double>, dim3(n), dim3(256), 0, (hipStream_t)stream, adj, r, bits, vals);
  GH_LAUNCH_CHECK();
  return 0;
}
extern "C" int gh_adj_pack_f32(const float* adj, int n, int r, uint64_t* bits, float* vals, gh_stream_t stream) {
  GH_REQUIRE(r > 0 && r <= MAX_R, "adj_pack: r=%d not in [1,%d]", r, MAX_R);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(adj_pack_kernel<float>, dim3(n), dim3(256), 0, (hipStream_t)stream, adj, r, bits, vals);
  GH_LAUNCH_CHECK();
  return 0;
}

extern "C" int gh_ref_depad(const int64_t* counts, int b, int n_max, int r, const void* ids, int ids_i64, const double* adj,
                            int32_t* d_ids, uint64_t* bits, float* vals, float* dinv, int32_t* n_nodes, int64_t* stats, int force_vals,
                            gh_stream_t stream) {
  GH_REQUIRE(r > 0 && r <= MAX_R, "ref_depad: r=%d not in [1,%d]", r, MAX_R);
  GH_REQUIRE(b >= 0 && n_max > 0 && counts && ids && adj && d_ids && bits && vals && dinv && n_nodes && stats, "ref_depad: bad arguments");
  GH_CHECK_HIP(hipMemsetAsync(stats, 0, 5 * sizeof(int64_t), (hipStream_t)stream));
  if (b == 0) return 0;
  if (ids_i64)
    hipLaunchKernelGGL(ref_depad_kernel<int64_t>, dim3(b * n_max), dim3(256), 0, (hipStream_t)stream, counts, b, n_max, r,
                       (const int64_t*)ids, adj, d_ids, bits, vals, dinv, n_nodes, (unsigned long long*)stats, force_vals);
  else
    hipLaunchKernelGGL(ref_depad_kernel<int32_t>, dim3(b * n_max), dim3(256), 0, (hipStream_t)stream, counts, b, n_max, r,
                       (const int32_t*)ids, adj, d_ids, bits, vals, dinv, n_nodes, (unsigned long long*)stats, force_vals);
  GH_LAUNCH_CHECK();
  return 0;
}

extern "C" int gh_adj_unpack(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep, int n,
                             int r, float* adj, gh_stream_t stream) {
  GH_REQUIRE(r > 0 && r <= MAX_R, "adj_unpack: r=%d not in [1,%d]", r, MAX_R);
  GH_REQUIRE(vals || dinv, "adj_unpack: need dinv or vals");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(adj_unpack_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, bits, dinv, vals, keep, r, adj);
  GH_LAUNCH_CHECK();
  return 0;
}

extern "C" int gh_spmm(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                       const int32_t* goff, int m_real, const float* x, float* y, int n, int r, int h, int transpose,
                       int accumulate, gh_stream_t stream) {
  if (n <= 0) return 0;
  return launch_spmm(bits, dinv, vals, keep, goff, m_real, x, y, n, r, h, transpose, accumulate, (hipStream_t)stream, 0);
}

extern "C" int gh_spmm_bf16(const uint64_t* bits, const float* dinv, const float* vals, const uint64_t* keep,
                            const int32_t* goff, int m_real, const void* x16, void* y16, int n, int r, int h, int transpose,
                            int accumulate, gh_stream_t stream) {
  if (n <= 0) return 0;
  GH_REQUIRE(h % 8 == 0, "spmm_bf16: h must be a multiple of 8");
  return launch_spmm(bits, dinv, vals, keep, goff, m_real, (const float*)x16, (float*)y16, n, r, h, transpose, accumulate, (hipStream_t)stream, 1);
}

extern "C" int gh_scorer_gsl(const uint64_t* bits, const float* dinv, const float* vals, const int32_t* goff,
                             int pads_collapsed, const float* feat, const float* score_x, const float* w_p, const float* gate, int n, int r, int h, int k, float* score,
                             uint64_t* keep, float drop_p, uint32_t drop_seed, gh_stream_t stream) {
  return scorer_gsl_impl(bits, dinv, vals, goff, pads_collapsed, feat, score_x, 1, 0, w_p, gate, n, r, h, k, score, keep, drop_p, drop_seed,
                         (hipStream_t)stream);
}

int gh::scorer_gsl_impl(const uint64_t* bits, const float* dinv, const float* vals, const int32_t* goff, int pads_collapsed, const float* feat,
                        const float* score_x, int score_parts, long long score_stride, const float* w_p, const float* gate, int n, int r, int h,
                        int k, float* score, uint64_t* keep, float drop_p, uint32_t drop_seed, hipStream_t stream) {
  GH_REQUIRE(r > 0 && r <= MAX_R, "scorer_gsl: r=%d not in [1,%d]", r, MAX_R);
  GH_REQUIRE(score_parts >= 1 && score_parts <= 16, "scorer_gsl: %d score partials", score_parts);
  GH_REQUIRE(vals || dinv, "scorer_gsl: need dinv or vals");
  if (n <= 0) return 0;
  prof_begin((hipStream_t)stream, PROF_SCORER_GSL);
  GH_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "scorer_gsl: dropout p=%f not in [0,1)", drop_p);
  GH_REQUIRE(!(pads_collapsed && drop_p > 0.f), "scorer_gsl: collapsed padding rows are an evaluation-mode layout (no dropout)");
  GH_REQUIRE(drop_p <= 0.f || (long long)n * r * (long long)h < (1LL << 32), "scorer_gsl: %d rows x %d columns exceed the dropout mask's 32-bit element index", n * r, h);
  const double th = (double)drop_p * 4294967296.0;
  const unsigned thresh = drop_p > 0.f ? (th >= 4294967295.0 ? 4294967295u : (unsigned)th) : 0u;
  GH_REQUIRE((feat != nullptr) != (score_x != nullptr), "scorer_gsl: exactly one of feat / score_x");
  hipLaunchKernelGGL(scorer_gsl_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, bits, dinv, vals, goff, feat, score_x,
                     w_p, gate, r, h, k, (goff && pads_collapsed) ? 1 : 0, score, keep, thresh, 1.0f / (1.0f - drop_p), drop_seed,
                     score_parts, score_stride);
  prof_end(PROF_SCORER_GSL, (double)n * ((feat ? 4.0 * r * h : 4.0 * r) + 8.0 * r * words_for(r) + 8.0 * r + 8.0 * words_for(r)),
           (hipStream_t)stream);
  GH_LAUNCH_CHECK();
  return 0;
}

extern "C" int gh_gsl_topk(const float* score, int n, int r, int k, uint64_t* keep, gh_stream_t stream) {
  GH_REQUIRE(r > 0 && r <= MAX_R, "gsl_topk: r=%d not in [1,%d]", r, MAX_R);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gsl_topk_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, score, r, k, keep);
  GH_LAUNCH_CHECK();
  return 0;
}
