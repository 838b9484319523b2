// Streaming (HBM-bound) kernels around the GEMMs: gate-backward elementwise, column sums, masked
// softmax + weighted reduce of the concat attention (fwd/bwd), ragged claim<->evidence helpers,
// weight transpose, fused flat Adam.  Plus the library's error plumbing.
#include "../../include/get_hip.h"
#include "common.h"
#include <mutex>
#include <unordered_map>
#include "gemm.hip.h"
#include <stdarg.h>
#include <vector>

namespace gh {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------- event profiler
struct ProfRec { hipEvent_t a, b; int tag; double work; };
static bool g_prof_on = false;
static unsigned g_prof_mask = 0xFFFFFFFFu;     // bit t set: kernels with tag t are instrumented (gh_profile_select)
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static hipEvent_t g_prof_cur = nullptr;

static hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
bool prof_enabled() { return g_prof_on; }
void prof_begin(hipStream_t s, int tag) {
  g_prof_cur = nullptr;
  if (!g_prof_on || !((g_prof_mask >> tag) & 1u)) return;
  g_prof_cur = prof_event();
  if (g_prof_cur) (void)hipEventRecord(g_prof_cur, s);
}
void prof_end(int tag, double work, hipStream_t s) {
  if (!g_prof_on || !g_prof_cur) return;
  hipEvent_t b = prof_event();
  if (!b) return;
  (void)hipEventRecord(b, s);
  g_prof_recs.push_back(ProfRec{g_prof_cur, b, tag, work});
  g_prof_cur = nullptr;
}

// ---------------------------------------------------------------------------- transpose
__global__ void __launch_bounds__(256)
transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int rows, int cols) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    if (r < rows && c < cols) tile[j][tx] = w[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;      // output row = c, output col = r
    if (r < rows && c < cols) wt[(size_t)c * rows + r] = tile[tx][j];
  }
}

struct TransposeBatch { const float* src[32]; float* dst[32]; int rows[32]; int cols[32]; int tile0[33]; int n;
                        unsigned short* w16[32]; unsigned short* t16[32]; };      // optional bf16 twins of src and of src^T (gh_weights_refresh)
__device__ __forceinline__ unsigned short bf16_rne(float f) {
  unsigned u = __builtin_bit_cast(unsigned, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);      // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__global__ void __launch_bounds__(256)
transpose_batch_kernel(const TransposeBatch B) {
  __shared__ float tile[32][33];
  int m = 0;
  while (m + 1 < B.n && (int)blockIdx.x >= B.tile0[m + 1]) ++m;
  const int rows = B.rows[m], cols = B.cols[m];
  const int t = blockIdx.x - B.tile0[m], tx_n = (cols + 31) / 32;
  const int bx = (t % tx_n) * 32, by = (t / tx_n) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* w = B.src[m];
  float* wt = B.dst[m];
  unsigned short* w16 = B.w16[m];
  unsigned short* t16 = B.t16[m];
  // float4-shaped matrices (every weight matrix of the path but the 2-row head): one 16-byte access per thread and direction instead
  // of four 4-byte ones (configs[4]: 86 us per step for the 14 cell matrices, the 41 MB head layer and their bf16 twins)
  const bool vec = rows % 4 == 0 && cols % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(wt)) & 15) == 0 &&
                   ((reinterpret_cast<uintptr_t>(w16) | reinterpret_cast<uintptr_t>(t16)) & 7) == 0;
  if (vec) {
    const int q8 = threadIdx.x & 7, r8 = threadIdx.x >> 3;      // 32 rows x 8 float4 columns
    {
      const int r = by + r8, c = bx + 4 * q8;
      if (r < rows && c < cols) {
        const float4 v = *reinterpret_cast<const float4*>(w + (size_t)r * cols + c);
        tile[r8][4 * q8 + 0] = v.x; tile[r8][4 * q8 + 1] = v.y; tile[r8][4 * q8 + 2] = v.z; tile[r8][4 * q8 + 3] = v.w;
        if (w16) *reinterpret_cast<uint2*>(w16 + (size_t)r * cols + c) =
            make_uint2((unsigned)bf16_rne(v.x) | ((unsigned)bf16_rne(v.y) << 16), (unsigned)bf16_rne(v.z) | ((unsigned)bf16_rne(v.w) << 16));
      }
    }
    __syncthreads();
    {
      const int c = bx + r8, r = by + 4 * q8;      // output row = source column c, four consecutive source rows
      if (c < cols && r < rows) {
        const float4 v = make_float4(tile[4 * q8 + 0][r8], tile[4 * q8 + 1][r8], tile[4 * q8 + 2][r8], tile[4 * q8 + 3][r8]);
        if (wt) *reinterpret_cast<float4*>(wt + (size_t)c * rows + r) = v;
        if (t16) *reinterpret_cast<uint2*>(t16 + (size_t)c * rows + r) =
            make_uint2((unsigned)bf16_rne(v.x) | ((unsigned)bf16_rne(v.y) << 16), (unsigned)bf16_rne(v.z) | ((unsigned)bf16_rne(v.w) << 16));
      }
    }
    return;
  }
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    if (r < rows && c < cols) {
      const float v = w[(size_t)r * cols + c];
      tile[j][tx] = v;
      if (w16) w16[(size_t)r * cols + c] = bf16_rne(v);
    }
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;
    if (r < rows && c < cols) {
      const float v = tile[tx][j];
      if (wt) wt[(size_t)c * rows + r] = v;
      if (t16) t16[(size_t)c * rows + r] = bf16_rne(v);
    }
  }
}

// ---------------------------------------------------------------------------- GGNN gate backward (elementwise part)
// out = h z + xp (1 - z)  (wrapper.py:206):  dhp = g z (1-h^2) ; dzp = g (h-xp) z (1-z) ; dxp = g (1-z)
__global__ void __launch_bounds__(256)
gate_bwd_pre_kernel(const float4* __restrict__ g, const float4* __restrict__ z, const float4* __restrict__ hh,
                    const float4* __restrict__ xp, float4* __restrict__ dhp, float4* __restrict__ dzp,
                    float4* __restrict__ dxp, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 G = g[i], Z = z[i], Hh = hh[i], X = xp[i];
    float4 a, b, c;
#define GH_ONE(f)                                      \
    a.f = G.f * Z.f * (1.f - Hh.f * Hh.f);             \
    b.f = G.f * (Hh.f - X.f) * Z.f * (1.f - Z.f);      \
    c.f = G.f * (1.f - Z.f);
    GH_ONE(x) GH_ONE(y) GH_ONE(z) GH_ONE(w)
#undef GH_ONE
    dhp[i] = a; dzp[i] = b; dxp[i] = c;
  }
}
__global__ void __launch_bounds__(256)
gate_bwd_pre_scalar_kernel(const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ hh,
                           const float* __restrict__ xp, float* __restrict__ dhp, float* __restrict__ dzp,
                           float* __restrict__ dxp, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float G = g[i], Z = z[i], Hh = hh[i], X = xp[i];
    dhp[i] = G * Z * (1.f - Hh * Hh);
    dzp[i] = G * (Hh - X) * Z * (1.f - Z);
    dxp[i] = G * (1.f - Z);
  }
}

__device__ __forceinline__ float4 mbf4_to_f4(uint2 u) {
  return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                     __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
}
__device__ __forceinline__ unsigned mpack_bf2(float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// bf16 storage pipeline: g fp32, every other stream bf16 (4 elements = 8 bytes per item)
__global__ void __launch_bounds__(256)
gate_bwd_pre_bf16_kernel(const float4* __restrict__ g, const uint2* __restrict__ z, const uint2* __restrict__ hh,
                         const uint2* __restrict__ xp, uint2* __restrict__ dhp, uint2* __restrict__ dzp,
                         uint2* __restrict__ dxp, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 G = g[i], Z = mbf4_to_f4(z[i]), Hh = mbf4_to_f4(hh[i]), X = mbf4_to_f4(xp[i]);
    float4 a, b, c;
#define GH_ONE(f)                                      \
    a.f = G.f * Z.f * (1.f - Hh.f * Hh.f);             \
    b.f = G.f * (Hh.f - X.f) * Z.f * (1.f - Z.f);      \
    c.f = G.f * (1.f - Z.f);
    GH_ONE(x) GH_ONE(y) GH_ONE(z) GH_ONE(w)
#undef GH_ONE
    dhp[i] = make_uint2(mpack_bf2(a.x, a.y), mpack_bf2(a.z, a.w));
    dzp[i] = make_uint2(mpack_bf2(b.x, b.y), mpack_bf2(b.z, b.w));
    dxp[i] = make_uint2(mpack_bf2(c.x, c.y), mpack_bf2(c.z, c.w));
  }
}

int launch_gate_bwd_pre(const float* g, const float* z, const float* hh, const float* xp, float* dhp, float* dzp,
                        float* dxp, size_t count, hipStream_t s, int bf16) {
  if (count == 0) return 0;
  if (bf16) {
    GH_REQUIRE(count % 4 == 0, "gate_bwd_pre: the bf16 variant needs a multiple of 4 elements");
    const size_t n4 = count / 4;
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    prof_begin(s, PROF_GATE_BWD_PRE);
    hipLaunchKernelGGL(gate_bwd_pre_bf16_kernel, dim3(grid), dim3(256), 0, s, (const float4*)g, (const uint2*)z, (const uint2*)hh,
                       (const uint2*)xp, (uint2*)dhp, (uint2*)dzp, (uint2*)dxp, n4);
    prof_end(PROF_GATE_BWD_PRE, (4.0 + 6.0 * 2.0) * (double)count, s);
    GH_LAUNCH_CHECK();
    return 0;
  }
  const uintptr_t al = (uintptr_t)g | (uintptr_t)z | (uintptr_t)hh | (uintptr_t)xp | (uintptr_t)dhp | (uintptr_t)dzp | (uintptr_t)dxp;
  if (count % 4 == 0 && (al & 15) == 0) {
    const size_t n4 = count / 4;
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    const int ptag = count < ((size_t)1 << 21) ? PROF_FEW_ROWS : PROF_GATE_BWD_PRE;      // (claim-side cells: 960 x 300)
    prof_begin(s, ptag);
    hipLaunchKernelGGL(gate_bwd_pre_kernel, dim3(grid), dim3(256), 0, s, (const float4*)g, (const float4*)z,
                       (const float4*)hh, (const float4*)xp, (float4*)dhp, (float4*)dzp, (float4*)dxp, n4);
    prof_end(ptag, 7.0 * 4.0 * (double)count, s);
  } else {
    const int grid = (int)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    hipLaunchKernelGGL(gate_bwd_pre_scalar_kernel, dim3(grid), dim3(256), 0, s, g, z, hh, xp, dhp, dzp, dxp, count);
  }
  GH_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------- column sums (bias gradients), += via atomics
constexpr int CS_ROWS = 256;
__global__ void __launch_bounds__(256)
colsum_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
              float* __restrict__ oa, float* __restrict__ ob, float* __restrict__ oc, float* __restrict__ oa2,
              float* __restrict__ ob2, float* __restrict__ oc2, int m, int h) {
  const int r0 = blockIdx.x * CS_ROWS, r1 = min(m, r0 + CS_ROWS);
  for (int col = threadIdx.x; col < h; col += 256) {
    float sa = 0.f, sb = 0.f, sc = 0.f;
    for (int r = r0; r < r1; ++r) {
      sa += a[(size_t)r * h + col];
      if (b) sb += b[(size_t)r * h + col];
      if (c) sc += c[(size_t)r * h + col];
    }
    atomicAdd(oa + col, sa);
    if (b) atomicAdd(ob + col, sb);
    if (c) atomicAdd(oc + col, sc);
    if (oa2) atomicAdd(oa2 + col, sa);
    if (b && ob2) atomicAdd(ob2 + col, sb);
    if (c && oc2) atomicAdd(oc2 + col, sc);
  }
}
// workspace variant: float4 columns x RL row lanes per block, partial sums to scratch, one final pass.
constexpr int CS2_ROWS = 128;      // rows per block of activation-sized inputs; few-row inputs (claim cell at h = 768: 960 rows x 192 float4
                                   // columns, ONE row lane per block) take 16 -- eight blocks walked 128 rows each in 125 us
__global__ void colsum_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                      float* __restrict__ ws, int m, int h, int RL, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  float4* red = reinterpret_cast<float4*>(dsm);          // [3][RL][h/4]
  const int n4 = h / 4;
  const int c4 = threadIdx.x % n4, rl = threadIdx.x / n4;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(m, r0 + rows_per_block);
  float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa, sc = sa;
  if (rl < RL) {
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += RL) {
      const size_t o = (size_t)r * n4 + c4;
      const float4 va = reinterpret_cast<const float4*>(a)[o];
      sa.x += va.x; sa.y += va.y; sa.z += va.z; sa.w += va.w;
      if (b) { const float4 v = reinterpret_cast<const float4*>(b)[o]; sb.x += v.x; sb.y += v.y; sb.z += v.z; sb.w += v.w; }
      if (c) { const float4 v = reinterpret_cast<const float4*>(c)[o]; sc.x += v.x; sc.y += v.y; sc.z += v.z; sc.w += v.w; }
    }
    red[(0 * RL + rl) * n4 + c4] = sa;
    red[(1 * RL + rl) * n4 + c4] = sb;
    red[(2 * RL + rl) * n4 + c4] = sc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * n4; i += blockDim.x) {
    const int arr = i / n4, cc = i % n4;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < RL; ++k) {
      const float4 v = red[(arr * RL + k) * n4 + cc];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    reinterpret_cast<float4*>(ws)[((size_t)blockIdx.x * 3 + arr) * n4 + cc] = t;
  }
}
// 64 columns x 4 partial-row lanes per block; LDS tree over the lanes
__global__ void __launch_bounds__(256)
colsum_final_kernel(const float* __restrict__ ws, int nblk, int h, float* __restrict__ oa, float* __restrict__ ob,
                    float* __restrict__ oc, float* __restrict__ oa2, float* __restrict__ ob2, float* __restrict__ oc2) {
  __shared__ float red[4][64];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), kl = threadIdx.x >> 6;
  float acc = 0.f;
  if (i < 3 * h) {
    const int arr = i / h, col = i % h;
#pragma unroll 4
    for (int k = kl; k < nblk; k += 4) acc += ws[((size_t)k * 3 + arr) * h + col];
  }
  red[kl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (kl == 0 && i < 3 * h) {
    const int arr = i / h, col = i % h;
    float* out = arr == 0 ? oa : (arr == 1 ? ob : oc);
    float* out2 = arr == 0 ? oa2 : (arr == 1 ? ob2 : oc2);
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (out) out[col] += v;
    if (out2) out2[col] += v;     // the same sum feeds a second bias (b?0 and b?1 share their gradient)
  }
}


int launch_colsum3(const float* a, const float* b, const float* c, float* oa, float* ob, float* oc, int m, int h,
                   hipStream_t s, float* oa2, float* ob2, float* oc2) {
  if (m <= 0) return 0;
  const uintptr_t al = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c;
  const int rpb = m <= 8192 ? 16 : CS2_ROWS;
  const int nblk = (m + rpb - 1) / rpb;
  const size_t need = (size_t)nblk * 3 * h * sizeof(float);
  const double bytes = 4.0 * (double)m * h * (1 + (b != nullptr) + (c != nullptr));
  const Workspace wsp = workspace_for(s);
  float* const g_cs_ws = wsp.cs;
  if (g_cs_ws && need <= wsp.cs_bytes && h % 4 == 0 && h / 4 <= 256 && (al & 15) == 0) {
    const int n4 = h / 4;
    const int RL = (256 / n4) > 0 ? (256 / n4) : 1;
    const int threads = ((n4 * RL + 63) / 64) * 64;
    prof_begin(s, PROF_COLSUM);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(threads), (size_t)3 * RL * h * sizeof(float), s, a, b, c,
                       g_cs_ws, m, h, RL, rpb);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((3 * h + 63) / 64), dim3(256), 0, s, g_cs_ws, nblk, h, oa, ob, oc, oa2, ob2, oc2);
    prof_end(PROF_COLSUM, bytes, s);
    GH_LAUNCH_CHECK();
    return 0;
  }
  prof_begin(s, PROF_COLSUM);
  hipLaunchKernelGGL(colsum_kernel, dim3((m + CS_ROWS - 1) / CS_ROWS), dim3(256), 0, s, a, b, c, oa, ob, oc, oa2, ob2, oc2, m, h);
  prof_end(PROF_COLSUM, bytes, s);
  GH_LAUNCH_CHECK();
  return 0;
}
int launch_colsum(const float* a, float* oa, int m, int h, hipStream_t s) {
  return launch_colsum3(a, nullptr, nullptr, oa, nullptr, nullptr, m, h, s, nullptr, nullptr, nullptr);
}

// ---------------------------------------------------------------------------- row gather (embedding rows for the dW_proj GEMM)
// dst[r][:] = dropout(table[ids ? ids[r] : r][:]) -- the operand of the dW_proj GEMM, with the forward's mask
__global__ void __launch_bounds__(256)
gather_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids, float* __restrict__ dst, int m,
                   int d, unsigned drop_thresh, float drop_scale, unsigned drop_seed) {
  const int per = (d + 3) / 4;
  if (d % 4 == 0 && ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    // four independent id -> row chains per thread and trip (as gather_rows_bf16_kernel: one chain at a time left the kernel parked
    // on its loads 82 % of the time)
    const size_t total = (size_t)m * per, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t it0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it0 < total; it0 += 4 * stride) {
      int r[4], c[4], src[4];
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t it = it0 + k * stride < total ? it0 + k * stride : it0;
        r[k] = (int)(it / per); c[k] = 4 * (int)(it % per);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) src[k] = ids ? ids[r[k]] : r[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(table + (size_t)src[k] * d + c[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (it0 + k * stride >= total) break;
        float4 w = v[k];
        if (drop_thresh) w = drop4(w, drop_seed, (unsigned)r[k] * (unsigned)d + (unsigned)c[k], drop_thresh, drop_scale);
        *reinterpret_cast<float4*>(dst + (size_t)r[k] * d + c[k]) = w;
      }
    }
    return;
  }
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < (size_t)m * per; it += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(it / per), c = 4 * (int)(it % per);
    const float* s = table + (size_t)(ids ? ids[r] : r) * d + c;
    float* o = dst + (size_t)r * d + c;
    if (d % 4 == 0) {
      float4 v = *reinterpret_cast<const float4*>(s);
      if (drop_thresh) v = drop4(v, drop_seed, (unsigned)r * (unsigned)d + (unsigned)c, drop_thresh, drop_scale);
      *reinterpret_cast<float4*>(o) = v;
    } else {
      for (int k = 0; k < 4 && c + k < d; ++k) {
        float v = s[k];
        if (drop_thresh) v = drop_hash(drop_seed, (unsigned)r * (unsigned)d + (unsigned)(c + k)) >= drop_thresh ? v * drop_scale : 0.f;
        o[k] = v;
      }
    }
  }
}
// bf16 storage pipeline: table rows and dst hold bf16
__global__ void __launch_bounds__(256)
gather_rows_bf16_kernel(const unsigned short* __restrict__ table, const int32_t* __restrict__ ids, unsigned short* __restrict__ dst,
                        int m, int d, unsigned drop_thresh, float drop_scale, unsigned drop_seed) {
  if (d % 8 == 0 && ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {      // 16-byte items (8 values)
    // four items per thread and trip, every load of the trip ahead of the first store: an item is a dependent id -> row chain, and
    // one chain per thread at a time left the kernel latency-bound (configs[4], 96 000 x 768: 117 us for 2 x 147 MB)
    const int per8 = d / 8;
    const size_t total = (size_t)m * per8, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t it0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it0 < total; it0 += 4 * stride) {
      int r[4], c[4];
      uint4 u[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t it = it0 + k * stride < total ? it0 + k * stride : it0;      // (clamped duplicates are loaded, never stored)
        r[k] = (int)(it / per8); c[k] = 8 * (int)(it % per8);
      }
      int src[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) src[k] = ids ? ids[r[k]] : r[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = *reinterpret_cast<const uint4*>(table + (size_t)src[k] * d + c[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (it0 + k * stride >= total) break;
        float4 va = mbf4_to_f4(make_uint2(u[k].x, u[k].y)), vb = mbf4_to_f4(make_uint2(u[k].z, u[k].w));
        if (drop_thresh) {
          const unsigned idx = (unsigned)r[k] * (unsigned)d + (unsigned)c[k];
          va = drop4(va, drop_seed, idx, drop_thresh, drop_scale);
          vb = drop4(vb, drop_seed, idx + 4u, drop_thresh, drop_scale);
        }
        *reinterpret_cast<uint4*>(dst + (size_t)r[k] * d + c[k]) = make_uint4(mpack_bf2(va.x, va.y), mpack_bf2(va.z, va.w), mpack_bf2(vb.x, vb.y), mpack_bf2(vb.z, vb.w));
      }
    }
    return;
  }
  const int per = d / 4;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < (size_t)m * per; it += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(it / per), c = 4 * (int)(it % per);
    float4 v = mbf4_to_f4(*reinterpret_cast<const uint2*>(table + (size_t)(ids ? ids[r] : r) * d + c));
    if (drop_thresh) v = drop4(v, drop_seed, (unsigned)r * (unsigned)d + (unsigned)c, drop_thresh, drop_scale);
    *reinterpret_cast<uint2*>(dst + (size_t)r * d + c) = make_uint2(mpack_bf2(v.x, v.y), mpack_bf2(v.z, v.w));
  }
}

int launch_gather_rows(const float* table, const int32_t* ids, float* dst, int m, int d, hipStream_t s, float drop_p,
                       unsigned drop_seed, int bf16) {
  if (m <= 0) return 0;
  GH_REQUIRE(drop_p <= 0.f || (long long)m * (long long)d < (1LL << 32), "gather_rows: %d rows x %d columns exceed the dropout mask's 32-bit element index", m, d);
  if (bf16) {
    GH_REQUIRE(d % 4 == 0, "gather_rows: the bf16 variant needs d %% 4 == 0");
    const size_t n = (size_t)m * (d / 4);
    const double th = (double)drop_p * 4294967296.0;
    const unsigned thresh = drop_p > 0.f ? (th >= 4294967295.0 ? 4294967295u : (unsigned)th) : 0u;
    hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s,
                       (const unsigned short*)table, ids, (unsigned short*)dst, m, d, thresh, 1.0f / (1.0f - drop_p), drop_seed);
    GH_LAUNCH_CHECK();
    return 0;
  }
  const size_t n = (size_t)m * ((d + 3) / 4);
  const double th = (double)drop_p * 4294967296.0;
  const unsigned thresh = drop_p > 0.f ? (th >= 4294967295.0 ? 4294967295u : (unsigned)th) : 0u;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s,
                     table, ids, dst, m, d, thresh, 1.0f / (1.0f - drop_p), drop_seed);
  GH_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------- concat attention: masked softmax + weighted reduce
// bf16 storage mode (R16): `right` is the producing cell's bf16 output (four values per 8-byte load, widened at the point of use --
// the loads stay in flight as raw words), so that no fp32 copy of that output has to exist
template <bool R16> struct RightVec { typedef float4 T; };
template <> struct RightVec<true> { typedef uint2 T; };
__device__ __forceinline__ float4 right_widen(const float4 v) { return v; }
__device__ __forceinline__ float4 right_widen(const uint2 v) {
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
// e [b][l][C] -> weights = softmax over l (two_branches_attention.py:142-146), attended[b][d][c] = sum_l right[b][l][d] w[l][c] (:147)
template <int CT, bool R16 = false>     // compile-time bound on the number of heads (accumulator registers); R16: right holds bf16
__global__ void __launch_bounds__(256, 4)
att_softmax_fwd_kernel(float* __restrict__ e, const float* __restrict__ mask, const float* __restrict__ right,
                       const int32_t* __restrict__ goff, int Lmax, int Dr, int C, float* __restrict__ weights,
                       float* __restrict__ attended, const float* __restrict__ e_parts, int n_parts, long long part_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  float* ws = reinterpret_cast<float*>(dsm);   // [Lmax][C]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // node-compact layout: pair b owns rows [goff[b], goff[b+1]) of e / mask / right / weights
  const int row0 = goff ? goff[b] : b * Lmax;
  const int L = goff ? goff[b + 1] - row0 : Lmax;
  const float* mb = mask + (size_t)row0;
  const int NT = blockDim.x, NWV = NT >> 6;
  // The right tile is what the kernel streams (L rows x Dr floats per pair): the first batch of this wave's rows is
  // requested BEFORE the softmax (its latency hides behind the exp / shuffle work), and every later batch before the
  // previous one is consumed -- RB rows = RB KB per wave in flight instead of the one-row-at-a-time walk that kept the
  // kernel latency-bound at 1.9 TB/s.
  // A workgroup takes 128 float4 columns (two per lane): at Dr = 300 that is the WHOLE row -- the 64-column slabs
  // left a second workgroup per pair reading 176-byte runs (11 float4) whose 128-byte lines the first one fetches too.
  // All 960 pairs of the bench shape should be resident at once (4 workgroups per CU -> <= 128 VGPRs), and a wave should
  // need as few memory round trips as possible for its ~16 rows: RB = 6 rows x 2 chunks in flight per trip, single
  // buffered (the math between two trips is far shorter than a round trip, so a second buffer only costs registers).
  constexpr int RB = CT <= 2 ? 8 : CT <= 5 ? 6 : 4, NCH = 2;   // (8 x 2 chunks + 40 accumulators spill at 128 VGPRs)
  const int D4 = Dr / 4;
  int dcl[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) dcl[h] = min(blockIdx.y * (64 * NCH) + lane + 64 * h, D4 - 1);
  typedef typename RightVec<R16>::T RV;
  const RV* rb = R16 ? reinterpret_cast<const RV*>(reinterpret_cast<const unsigned short*>(right) + (size_t)row0 * Dr)
                     : reinterpret_cast<const RV*>(right + (size_t)row0 * Dr);
  RV zero4;
  __builtin_memset(&zero4, 0, sizeof(RV));
  // (unconditional loads from clamped rows / columns: a `cond ? load : 0` select makes the compiler route the load
  // through a flat pointer to a zero in scratch; clamped duplicates are loaded twice and never consumed)
  RV rv[RB][NCH];
  if (L > 0) {
#pragma unroll
    for (int u = 0; u < RB; ++u)
#pragma unroll
      for (int h = 0; h < NCH; ++h) rv[u][h] = rb[(size_t)min(wave + u * NWV, L - 1) * D4 + dcl[h]];
  } else {
#pragma unroll
    for (int u = 0; u < RB; ++u)
#pragma unroll
      for (int h = 0; h < NCH; ++h) rv[u][h] = zero4;
  }
  // scores staged in LDS (read twice below).  n_parts > 0: the producing GEMM left one partial per column block of the hidden
  // layer (wide rows, h = 768): summed here in block order, and the sum is written back as the layer's `e` output
  float* eb = ws + (size_t)Lmax * C + (size_t)NWV * 64 * (4 * C + 1);      // [Lmax][C], behind the weights and the reduce partials
  for (int i = tid; i < L * C; i += NT) {
    float v;
    if (n_parts > 0) {
      v = 0.f;
      for (int p = 0; p < n_parts; ++p) v += e_parts[(size_t)p * part_stride + (size_t)row0 * C + i];
      if (blockIdx.y == 0) e[(size_t)row0 * C + i] = v;
    } else {
      v = e[(size_t)row0 * C + i];
    }
    eb[i] = v;
  }
  __syncthreads();
  for (int c = wave; c < C; c += NWV) {
    float mx = -INFINITY;
    for (int l = lane; l < L; l += 64)
      if (mb[l] != 0.f) mx = fmaxf(mx, eb[l * C + c]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float p = (mb[l] != 0.f) ? expf(eb[l * C + c] - mx) : 0.f;
      ws[l * C + c] = p;
      sum += p;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;          // all-masked row: 0/0 = NaN, as the reference's softmax of -inf
    for (int l = lane; l < L; l += 64) ws[l * C + c] = (sum > 0.f) ? ws[l * C + c] * inv : NAN;
  }
  __syncthreads();
  if (blockIdx.y == 0)
    for (int i = tid; i < L * C; i += NT) weights[(size_t)row0 * C + i] = ws[i];
  // attended[d][c] = sum_l right[l][d] w[l][c]: the waves take every NWV-th row (16-byte coalesced reads), partial sums
  // meet in LDS, one column chunk after the other
  // [NWV][64 lanes][4 C + 1] floats.  The odd pitch keeps the 4 C partial stores of a wave conflict-free (at the pitch 4 C the
  // 64 lanes of a store sat 4 C floats apart: 32-way bank conflicts at C = 8, 4-way at C = 5 -- PMC on configs[4]:
  // SQ_LDS_BANK_CONFLICT 87 % of the kernel's LDS cycles) and the reduce below still reads consecutive words
  float* part = ws + Lmax * C;
  const int PP = 4 * C + 1;
  float acc[NCH][4][CT];
#pragma unroll
  for (int h = 0; h < NCH; ++h)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[h][k][c] = 0.f;
  for (int l0 = wave; l0 < L; l0 += RB * NWV) {
    if (l0 != wave) {
#pragma unroll
      for (int u = 0; u < RB; ++u)
#pragma unroll
        for (int h = 0; h < NCH; ++h) rv[u][h] = rb[(size_t)min(l0 + u * NWV, L - 1) * D4 + dcl[h]];
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int l = l0 + u * NWV;
      if (l < L) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
          if (c < C) {
            const float w = ws[l * C + c];
#pragma unroll
            for (int h = 0; h < NCH; ++h) {
              const float4 r4 = right_widen(rv[u][h]);
              acc[h][0][c] += r4.x * w; acc[h][1][c] += r4.y * w; acc[h][2][c] += r4.z * w; acc[h][3][c] += r4.w * w;
            }
          }
      }
    }
  }
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    if (h > 0) __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (c < C) part[(wave * 64 + lane) * PP + k * C + c] = acc[h][k][c];
    __syncthreads();
    for (int i = tid; i < 64 * 4 * C; i += NT) {
      const int ln = i / (4 * C), rem = i % (4 * C);     // rem = k*C + c -> output offset within the float4 column group
      const int dd = (blockIdx.y * (64 * NCH) + 64 * h + ln) * 4 + rem / C;
      if (dd < Dr) {
        float v = 0.f;
        for (int w = 0; w < NWV; ++w) v += part[(w * 64 + ln) * PP + rem];
        if (L == 0) v = NAN;                      // no rows at all: the reference's softmax over an all -inf column
        attended[((size_t)b * Dr + dd) * C + rem % C] = v;
      }
    }
  }
}

int launch_att_softmax_fwd(float* e, const float* mask, const float* right, const int32_t* goff, int m_real,
                           int b, int l, int dr, int heads, float* weights, float* attended, hipStream_t s,
                           const float* e_parts, int n_parts, long long part_stride, const void* right16) {
  GH_REQUIRE(right || right16, "att_softmax_fwd: no right operand");
  if (right16) {
    GH_REQUIRE(dr % 4 == 0 && (reinterpret_cast<uintptr_t>(right16) & 7) == 0, "att_softmax_fwd: bf16 right rows must be 8-byte shaped (dr=%d)", dr);
    right = reinterpret_cast<const float*>(right16);
  }
  GH_REQUIRE(dr % 4 == 0 && (right16 || (reinterpret_cast<uintptr_t>(right) & 15) == 0), "att_softmax_fwd: right rows must be float4-shaped (dr=%d)", dr);
  const int nthr = 256;      // (8 waves per pair measured no faster: the kernel is not parallelism-bound)
  const size_t lds = ((size_t)2 * l * heads + (nthr / 64) * 64 * (4 * heads + 1)) * 4;
  GH_REQUIRE(lds <= 64 * 1024, "att_softmax_fwd: sequence %d x heads %d too large", l, heads);
  prof_begin(s, b < PROF_FEW_GROUPS ? PROF_FEW_ROWS : PROF_ATT_SOFTMAX_FWD);
  GH_REQUIRE(heads >= 1 && heads <= 8, "att_softmax_fwd: %d heads (1..8 supported)", heads);
  const dim3 grid(b, (dr / 4 + 127) / 128);
  if (right16) {
    if (heads <= 2) hipLaunchKernelGGL((att_softmax_fwd_kernel<2, true>), grid, dim3(nthr), lds, s, e, mask, right, goff, l, dr, heads, weights, attended, e_parts, n_parts, part_stride);
    else if (heads <= 5) hipLaunchKernelGGL((att_softmax_fwd_kernel<5, true>), grid, dim3(nthr), lds, s, e, mask, right, goff, l, dr, heads, weights, attended, e_parts, n_parts, part_stride);
    else hipLaunchKernelGGL((att_softmax_fwd_kernel<8, true>), grid, dim3(nthr), lds, s, e, mask, right, goff, l, dr, heads, weights, attended, e_parts, n_parts, part_stride);
  }
  else if (heads <= 2) hipLaunchKernelGGL(att_softmax_fwd_kernel<2>, grid, dim3(nthr), lds, s, e, mask, right, goff, l, dr, heads, weights, attended, e_parts, n_parts, part_stride);
  else if (heads <= 5) hipLaunchKernelGGL(att_softmax_fwd_kernel<5>, grid, dim3(nthr), lds, s, e, mask, right, goff, l, dr, heads, weights, attended, e_parts, n_parts, part_stride);
  else hipLaunchKernelGGL(att_softmax_fwd_kernel<8>, grid, dim3(nthr), lds, s, e, mask, right, goff, l, dr, heads, weights, attended, e_parts, n_parts, part_stride);
  const double rows = goff ? (double)m_real : (double)b * l;
  prof_end(b < PROF_FEW_GROUPS ? PROF_FEW_ROWS : PROF_ATT_SOFTMAX_FWD, (right16 ? 2.0 : 4.0) * rows * dr + 4.0 * (2.0 * rows * heads + rows + (double)b * dr * heads), s);
  GH_LAUNCH_CHECK();
  return 0;
}

// Sum over the 64 lanes of a wave on the VALU's DPP path (row shifts + row broadcasts, gfx9 wave64): six VALU
// instructions, no LDS crossbar traffic -- __shfl_xor compiles to ds_bpermute_b32 (an LDS instruction plus a wait per
// step), which made the five head reductions per row the largest cost of the word-level softmax backward.  The total
// lands in lane 63 (returned to every lane through readlane).
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));   // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xe, true));   // row_shr:4, banks 1-3
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xc, true));   // row_shr:8, banks 2-3
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));   // row_bcast:15 -> rows 1, 3
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, true));   // row_bcast:31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// backward of the softmax/reduce: dw[l][c] = sum_d right[l][d] g_att[d][c] (+ g_w) ; de = w (dw - sum_l w dw) ;
// dright[l][d] = sum_c w[l][c] g_att[d][c]  (first contribution; the GEMM adds dpre W1r on top)
template <int CT>     // compile-time bound on the number of heads
__global__ void __launch_bounds__(512)
att_softmax_bwd_kernel(const float* __restrict__ right, const float* __restrict__ weights,
                       const float* __restrict__ g_att, const float* __restrict__ g_w,
                       const int32_t* __restrict__ goff, int Lmax, int Dr, int C,
                       float* __restrict__ de, float* __restrict__ dright) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  float* ga = reinterpret_cast<float*>(dsm);   // [C][Dr]: head-major, so a lane's four g values are one 16-byte read and
                                               // consecutive lanes hit consecutive banks ([Dr][C] was an 8-way conflict)
  float* ws = ga + (size_t)Dr * C;             // [Lmax][C]
  float* dw = ws + (size_t)Lmax * C;           // [Lmax][C]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = goff ? goff[b] : b * Lmax;
  const int L = goff ? goff[b + 1] - row0 : Lmax;
  const int NT = blockDim.x, NWV = NT >> 6;
  const int D4 = Dr / 4;
  constexpr int NCH = 2;                         // float4 columns per lane on the register path (Dr <= 512)
  const bool reg_path = D4 <= 64 * NCH;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int dcl[NCH];                                  // clamped column of this lane (loads stay unconditional)
#pragma unroll
  for (int h = 0; h < NCH; ++h) dcl[h] = min(lane + 64 * h, D4 - 1);
  // two rows per wave and trip; the first trip's rows are requested before g_att / the weights are staged
  float4 na[NCH], nb[NCH];
  if (reg_path && L > 0) {
    const float4* ra_ = reinterpret_cast<const float4*>(right + ((size_t)row0 + min(wave, L - 1)) * Dr);
    const float4* rb_ = reinterpret_cast<const float4*>(right + ((size_t)row0 + min(wave + NWV, L - 1)) * Dr);
#pragma unroll
    for (int h = 0; h < NCH; ++h) { na[h] = ra_[dcl[h]]; nb[h] = rb_[dcl[h]]; }
  }
  for (int i = tid; i < Dr * C; i += NT) ga[(i % C) * Dr + i / C] = g_att[(size_t)b * Dr * C + i];
  for (int i = tid; i < L * C; i += NT) ws[i] = weights[(size_t)row0 * C + i];
  __syncthreads();
  if (reg_path) {
    // word level (Dr = H): this lane's g_att columns are loop invariants ([CT][NCH] float4 out of LDS once), the row
    // dot products are reduced on the DPP path -- per row no LDS instruction is left but the C weight broadcasts
    float4 gq[CT][NCH];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        gq[c][h] = reinterpret_cast<const float4*>(ga + (size_t)min(c, C - 1) * Dr)[dcl[h]];
        if (c >= C || lane + 64 * h >= D4) gq[c][h] = zero4;          // clamped duplicates contribute nothing
      }
    for (int l = wave; l < L; l += 2 * NWV) {
      const int lb = l + NWV;
      const bool has_b = lb < L;
      float4 ca[NCH], cb[NCH];
#pragma unroll
      for (int h = 0; h < NCH; ++h) { ca[h] = na[h]; cb[h] = nb[h]; }
      if (l + 2 * NWV < L) {                     // next trip's rows in flight while this trip is consumed
        const float4* ra_ = reinterpret_cast<const float4*>(right + ((size_t)row0 + l + 2 * NWV) * Dr);
        const float4* rb_ = reinterpret_cast<const float4*>(right + ((size_t)row0 + min(l + 3 * NWV, L - 1)) * Dr);
#pragma unroll
        for (int h = 0; h < NCH; ++h) { na[h] = ra_[dcl[h]]; nb[h] = rb_[dcl[h]]; }
      }
      float4 aa[NCH], ab[NCH];
#pragma unroll
      for (int h = 0; h < NCH; ++h) { aa[h] = zero4; ab[h] = zero4; }
      float pa[CT], pb[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        pa[c] = 0.f; pb[c] = 0.f;
        if (c < C) {
          const float wa = ws[l * C + c], wb = ws[(has_b ? lb : l) * C + c];
#pragma unroll
          for (int h = 0; h < NCH; ++h) {
            const float4 q = gq[c][h];
            aa[h].x += wa * q.x; aa[h].y += wa * q.y; aa[h].z += wa * q.z; aa[h].w += wa * q.w;
            ab[h].x += wb * q.x; ab[h].y += wb * q.y; ab[h].z += wb * q.z; ab[h].w += wb * q.w;
            pa[c] += ca[h].x * q.x + ca[h].y * q.y + ca[h].z * q.z + ca[h].w * q.w;
            pb[c] += cb[h].x * q.x + cb[h].y * q.y + cb[h].z * q.z + cb[h].w * q.w;
          }
        }
      }
      float4* da_ = reinterpret_cast<float4*>(dright + ((size_t)row0 + l) * Dr);
      float4* db_ = reinterpret_cast<float4*>(dright + ((size_t)row0 + (has_b ? lb : l)) * Dr);
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        const int d4 = lane + 64 * h;
        if (d4 < D4) { da_[d4] = aa[h]; if (has_b) db_[d4] = ab[h]; }
      }
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (c < C) {
          const float va = wave_sum_dpp(pa[c]), vb = wave_sum_dpp(pb[c]);
          if (lane == 0) {
            dw[l * C + c] = va + (g_w ? g_w[((size_t)row0 + l) * C + c] : 0.f);
            if (has_b) dw[lb * C + c] = vb + (g_w ? g_w[((size_t)row0 + lb) * C + c] : 0.f);
          }
        }
    }
  } else {
    // wide rows (evidence level: Dr = H * heads + source width): two rows per wave in flight, columns walked in a loop
    for (int l = wave; l < L; l += 2 * NWV) {
      const int lb = l + NWV;
      const bool has_b = lb < L;
      const int lbc = has_b ? lb : l;
      float pa[CT], pb[CT], wa[CT], wb[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        pa[c] = 0.f; pb[c] = 0.f;
        wa[c] = (c < C) ? ws[l * C + c] : 0.f;
        wb[c] = (c < C) ? ws[lbc * C + c] : 0.f;
      }
      const float4* ra_ = reinterpret_cast<const float4*>(right + ((size_t)row0 + l) * Dr);
      const float4* rb_ = reinterpret_cast<const float4*>(right + ((size_t)row0 + lbc) * Dr);
      float4* da_ = reinterpret_cast<float4*>(dright + ((size_t)row0 + l) * Dr);
      float4* db_ = reinterpret_cast<float4*>(dright + ((size_t)row0 + lbc) * Dr);
      for (int d4 = lane; d4 < D4; d4 += 64) {
        const float4 rva = ra_[d4];
        const float4 rvb = rb_[d4];
        float4 aa = zero4, ab = zero4;
#pragma unroll
        for (int c = 0; c < CT; ++c)
          if (c < C) {
            const float4 gq = reinterpret_cast<const float4*>(ga + (size_t)c * Dr)[d4];
            aa.x += wa[c] * gq.x; aa.y += wa[c] * gq.y; aa.z += wa[c] * gq.z; aa.w += wa[c] * gq.w;
            ab.x += wb[c] * gq.x; ab.y += wb[c] * gq.y; ab.z += wb[c] * gq.z; ab.w += wb[c] * gq.w;
            pa[c] += rva.x * gq.x + rva.y * gq.y + rva.z * gq.z + rva.w * gq.w;
            pb[c] += rvb.x * gq.x + rvb.y * gq.y + rvb.z * gq.z + rvb.w * gq.w;
          }
        da_[d4] = aa;
        if (has_b) db_[d4] = ab;
      }
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (c < C) {
          const float va = wave_sum_dpp(pa[c]), vb = wave_sum_dpp(pb[c]);
          if (lane == 0) {
            dw[l * C + c] = va + (g_w ? g_w[((size_t)row0 + l) * C + c] : 0.f);
            if (has_b) dw[lb * C + c] = vb + (g_w ? g_w[((size_t)row0 + lb) * C + c] : 0.f);
          }
        }
    }
  }
  __syncthreads();
  for (int c = wave; c < C; c += NWV) {
    float sum = 0.f;
    for (int l = lane; l < L; l += 64) sum += ws[l * C + c] * dw[l * C + c];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    for (int l = lane; l < L; l += 64)
      de[((size_t)row0 + l) * C + c] = ws[l * C + c] * (dw[l * C + c] - sum);
  }
}


// Row-balanced first half of the same backward for the word level (Dr <= 512): one workgroup per PAIR left 960 workgroups
// of 158 VGPRs for 768 resident slots -- 1.25 rounds, 2.0 TB/s.  Here every WAVE owns an equal run of consecutive rows
// of the whole batch (pair boundaries are crossed by reloading the pair's g_att columns, 4 C floats per lane and chunk,
// contiguous in g_att's [Dr][C] layout), writes dright rows and the raw dw[row][c] = right[row] . g_att[pair][:, c] (+ g_w);
// the per-pair part of the softmax backward (de = w (dw - sum_l w dw)) moves into att_dpre_kernel's prologue, which
// stages those few values per pair anyway.  <= 128 VGPRs: 16 waves per CU, three rows in flight per wave.
template <int CT, bool R16 = false>      // exact number of heads; R16: right holds bf16
__global__ void __launch_bounds__(256, CT <= 5 ? 4 : 3)
att_rows_bwd_kernel(const float* __restrict__ right, const float* __restrict__ weights, const float* __restrict__ g_att,
                    const float* __restrict__ g_w, const int32_t* __restrict__ rowg, int Lmax, int Dr, int C, int M,
                    int rows_per_wave, float* __restrict__ dw_out, float* __restrict__ dright, int D4h) {
  // Rows wider than 512 floats (h = 768): blockIdx.y selects a column range of D4h float4 columns; every range writes its dright
  // columns and its PARTIAL dw (dw_out + range * M * C; att_dpre's prologue adds the ranges in order), g_w rides with range 0.
  constexpr int NCH = 2;
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int r0 = gw * rows_per_wave, r1 = min(M, r0 + rows_per_wave);
  if (r0 >= r1) return;
  const int D4 = Dr / 4;
  const int col0 = blockIdx.y * D4h;
  const int ncol = min(D4h, D4 - col0);           // float4 columns of this range
  dw_out += (size_t)blockIdx.y * (size_t)M * CT;
  if (blockIdx.y > 0) g_w = nullptr;
  int dcl[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) dcl[h] = col0 + min(lane + 64 * h, ncol - 1);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  typedef typename RightVec<R16>::T RV;
  const RV* rp = reinterpret_cast<const RV*>(right);
  // this wave's softmax weights and pair ids, staged once into its private LDS region: a per-row global load in the
  // loop (weights[l][c], rowg[l]) put one unhidden memory latency into every iteration
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  float* wl = reinterpret_cast<float*>(dsm) + (size_t)(threadIdx.x >> 6) * rows_per_wave * (2 * CT + 1);
  int* pl = reinterpret_cast<int*>(wl + (size_t)rows_per_wave * CT);
  float* dwl = wl + (size_t)rows_per_wave * (CT + 1);      // this wave's dw values: written out in one coalesced pass at the end
  const int nr = r1 - r0;
  for (int i = lane; i < nr * CT; i += 64) wl[i] = weights[(size_t)r0 * CT + i];
  for (int i = lane; i < nr; i += 64) pl[i] = rowg ? rowg[r0 + i] : (r0 + i) / Lmax;
  // three rows in flight (loads unconditional from clamped rows; duplicates are never consumed)
  RV ra[NCH], rb[NCH], rc[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    ra[h] = rp[(size_t)r0 * D4 + dcl[h]];
    rb[h] = rp[(size_t)min(r0 + 1, r1 - 1) * D4 + dcl[h]];
    rc[h] = rp[(size_t)min(r0 + 2, r1 - 1) * D4 + dcl[h]];
  }
  float4 gq[CT][NCH];
  int cur = -1;
  // one row: `buf` holds it (requested three rows ago); its slot is re-requested for row l + 3 right away.  The three slots
  // keep their roles (the loop is unrolled by three): rotating them through register moves would make every iteration wait
  // for ALL outstanding loads -- a move out of a register that a load is still filling waits for that load.
  auto step = [&](int l, RV (&buf)[NCH]) __attribute__((always_inline)) {
    if (l >= r1) return;
    const int pair = pl[l - r0];
    if (pair != cur) {          // wave-uniform: this pair's g_att columns, 4 C consecutive floats per lane and chunk
      cur = pair;
      const float4* gp = reinterpret_cast<const float4*>(g_att + (size_t)pair * Dr * CT);
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        float G[4 * CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          const float4 v = gp[(size_t)dcl[h] * CT + j];
          G[4 * j] = v.x; G[4 * j + 1] = v.y; G[4 * j + 2] = v.z; G[4 * j + 3] = v.w;
        }
        const bool live = lane + 64 * h < ncol;
#pragma unroll
        for (int c = 0; c < CT; ++c)
          // element (d = 4 d4 + k, c) sits at G[k * C + c] (the kernel is instantiated for the exact head count: C == CT)
          gq[c][h] = live ? make_float4(G[0 * CT + c], G[1 * CT + c], G[2 * CT + c], G[3 * CT + c]) : zero4;
      }
    }
    float4 cu[NCH];
#pragma unroll
    for (int h = 0; h < NCH; ++h) cu[h] = right_widen(buf[h]);
    if (l + 3 < r1) {
#pragma unroll
      for (int h = 0; h < NCH; ++h) buf[h] = rp[(size_t)(l + 3) * D4 + dcl[h]];
    }
    float4 aa[NCH];
#pragma unroll
    for (int h = 0; h < NCH; ++h) aa[h] = zero4;
    float pa[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      pa[c] = 0.f;
      const float wv = wl[(l - r0) * CT + c];
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        const float4 q = gq[c][h];
        aa[h].x += wv * q.x; aa[h].y += wv * q.y; aa[h].z += wv * q.z; aa[h].w += wv * q.w;
        pa[c] += cu[h].x * q.x + cu[h].y * q.y + cu[h].z * q.z + cu[h].w * q.w;
      }
    }
    float4* dp = reinterpret_cast<float4*>(dright + (size_t)l * Dr);
#pragma unroll
    for (int h = 0; h < NCH; ++h)
      if (lane + 64 * h < ncol) dp[col0 + lane + 64 * h] = aa[h];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const float v = wave_sum_dpp(pa[c]);
      if (lane == 0) dwl[(l - r0) * CT + c] = v;
    }
  };
  for (int l = r0; l < r1; l += 3) {
    step(l, ra);
    step(l + 1, rb);
    step(l + 2, rc);
  }
  if (g_w) {
    for (int i = lane; i < nr * CT; i += 64) dw_out[(size_t)r0 * CT + i] = dwl[i] + g_w[(size_t)r0 * CT + i];
  } else {
    for (int i = lane; i < nr * CT; i += 64) dw_out[(size_t)r0 * CT + i] = dwl[i];
  }
}
int launch_att_softmax_bwd(const float* right, const float* weights, const float* g_att, const float* g_w,
                           const int32_t* goff, int m_real, int b, int l, int dr, int heads, float* de, float* dright,
                           hipStream_t s, const int32_t* rowg, float* dw_tmp, int* dw_written, const void* right16) {
  GH_REQUIRE(right || right16, "att_softmax_bwd: no right operand");
  GH_REQUIRE(dr % 4 == 0 && (reinterpret_cast<uintptr_t>(right) & 15) == 0 && (reinterpret_cast<uintptr_t>(right16) & 7) == 0 &&
             (reinterpret_cast<uintptr_t>(dright) & 15) == 0, "att_softmax_bwd: right rows must be float4-shaped (dr=%d)", dr);
  if (dw_written) *dw_written = 0;
  // many pairs, rows that fit two float4 chunks per lane, 16-byte aligned g_att rows: the row-balanced kernel; de is then
  // finished by att_dpre's prologue (launch_att_dpre with dw_in)
  // (rows wider than 512 floats: ceil(dr / 512) column ranges, each a grid row of its own; *dw_written = the number of ranges.
  //  Up to 16 ranges: the evidence level at h = 768 has 6272-float rows over only 32 claims -- the per-pair kernel below, one workgroup
  //  per claim, streamed 1.5 MB per workgroup in 90 us)
  if (dw_tmp && dw_written && dr / 4 <= 16 * 128 && (goff == nullptr || rowg != nullptr) &&
      (reinterpret_cast<uintptr_t>(g_att) & 15) == 0 && heads >= 1 && heads <= 8) {
    const int M = goff ? m_real : b * l;
    if (M <= 0) return 0;
    const int waves = 256 * 4 * 4;                       // 16 waves per CU
    int rpw = (M + waves - 1) / waves < 4 ? 4 : (M + waves - 1) / waves;
    if (rpw > 256) rpw = 256;                            // (LDS staging of a wave's weights: more waves instead of longer runs)
    const int nwg = ((M + rpw - 1) / rpw + 3) / 4;
    const size_t lds_rows = (size_t)4 * rpw * (2 * heads + 1) * 4;
    const int ptag = b < PROF_FEW_GROUPS ? PROF_FEW_ROWS : PROF_ATT_SOFTMAX_BWD;
    prof_begin(s, ptag);
    static const void* const fns[8] = {(const void*)att_rows_bwd_kernel<1>, (const void*)att_rows_bwd_kernel<2>,
                                       (const void*)att_rows_bwd_kernel<3>, (const void*)att_rows_bwd_kernel<4>,
                                       (const void*)att_rows_bwd_kernel<5>, (const void*)att_rows_bwd_kernel<6>,
                                       (const void*)att_rows_bwd_kernel<7>, (const void*)att_rows_bwd_kernel<8>};
    static const void* const fns16[8] = {(const void*)att_rows_bwd_kernel<1, true>, (const void*)att_rows_bwd_kernel<2, true>,
                                         (const void*)att_rows_bwd_kernel<3, true>, (const void*)att_rows_bwd_kernel<4, true>,
                                         (const void*)att_rows_bwd_kernel<5, true>, (const void*)att_rows_bwd_kernel<6, true>,
                                         (const void*)att_rows_bwd_kernel<7, true>, (const void*)att_rows_bwd_kernel<8, true>};
    const void* fn = right16 ? fns16[heads - 1] : fns[heads - 1];
    if (right16) right = reinterpret_cast<const float*>(right16);
    int Mv = M, rpwv = rpw;
    const int ranges = (dr / 4 + 127) / 128;
    int d4h = (dr / 4 + ranges - 1) / ranges;
    void* args[] = {(void*)&right, (void*)&weights, (void*)&g_att, (void*)&g_w, (void*)&rowg, (void*)&l, (void*)&dr, (void*)&heads,
                    (void*)&Mv, (void*)&rpwv, (void*)&dw_tmp, (void*)&dright, (void*)&d4h};
    (void)hipLaunchKernel(fn, dim3(nwg, ranges), dim3(256), args, lds_rows, s);
    prof_end(ptag, (right16 ? 6.0 : 8.0) * M * dr + 4.0 * (3.0 * M * heads + (double)b * dr * heads), s);
    GH_LAUNCH_CHECK();
    *dw_written = ranges;
    return 0;
  }
  GH_REQUIRE(right, "att_softmax_bwd: the per-pair kernel (few pairs / no row map) reads an fp32 right operand");
  const size_t lds = ((size_t)dr * heads + 2 * (size_t)l * heads) * 4;
  GH_REQUIRE(lds <= 160 * 1024, "att_softmax_bwd: %zu B of LDS needed", lds);
  GH_REQUIRE(heads >= 1 && heads <= 8, "att_softmax_bwd: %d heads (1..8 supported)", heads);
  const void* fn = heads <= 2 ? (const void*)att_softmax_bwd_kernel<2> : heads <= 5 ? (const void*)att_softmax_bwd_kernel<5>
                                                                                   : (const void*)att_softmax_bwd_kernel<8>;
  static bool attr[3] = {false, false, false};
  const int ai = heads <= 2 ? 0 : heads <= 5 ? 1 : 2;
  if (!attr[ai] && lds > 64 * 1024) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[ai] = true; }
  prof_begin(s, b < PROF_FEW_GROUPS ? PROF_FEW_ROWS : PROF_ATT_SOFTMAX_BWD);
  // few pairs (evidence level: one workgroup per claim): eight waves share the rows of a pair
  void* args[] = {(void*)&right, (void*)&weights, (void*)&g_att, (void*)&g_w, (void*)&goff, (void*)&l, (void*)&dr, (void*)&heads,
                  (void*)&de, (void*)&dright};
  (void)hipLaunchKernel(fn, dim3(b), dim3(b < 512 ? 512 : 256), args, lds, s);
  const double rows = goff ? (double)m_real : (double)b * l;
  prof_end(b < PROF_FEW_GROUPS ? PROF_FEW_ROWS : PROF_ATT_SOFTMAX_BWD, 4.0 * (2.0 * rows * dr + 3.0 * rows * heads + (double)b * dr * heads), s);
  GH_LAUNCH_CHECK();
  return 0;
}

// dpre[m][n] = (sum_c de[m][c] w2[c][n]) (1 - t[m][n]^2) ; du[b][n] = sum_l dpre[b*L+l][n] ;
// dw2 partial [b][c][n] = sum_l de[m][c] t[m][n]   (summed over b by reduce_partials afterwards).
// One workgroup per (pair, column slab of S4 float4): threads = (float4 column of the slab, row lane); row lanes take
// every RL-th row.  Slabs triple the number of workgroups (2880 at the bench shape) and cut a thread's serial row walk
// to a third -- the one-workgroup-per-pair version was latency-bound at 2.0 TB/s.
__global__ void att_dpre_kernel(const float* __restrict__ de, const float* __restrict__ w2, const float* __restrict__ t,
                                const int32_t* __restrict__ goff, int Lmax, int Ha, int C, int RL, int S4,
                                float* __restrict__ dpre, float* __restrict__ du, float* __restrict__ dw2_part,
                                const float* __restrict__ dw_in, const float* __restrict__ wts, float* __restrict__ de_out,
                                unsigned short* __restrict__ dpre16,        // dpre16 != NULL: dpre is written THERE as bf16 (RNE) instead
                                long long dw_range_stride, int dw_ranges) { // dw_ranges > 1: dw_in holds that many column-range partials, dw_range_stride floats apart
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  float4* red = reinterpret_cast<float4*>(dsm);           // [RL][1 + C][S4]
  const int b = blockIdx.x;
  const int row0 = goff ? goff[b] : b * Lmax;
  const int L = goff ? goff[b + 1] - row0 : Lmax;
  const int n4 = Ha / 4;
  float* des = reinterpret_cast<float*>(red + (size_t)RL * (1 + C) * S4);       // [L][C]: read once, not once per thread and row
  const int s0 = blockIdx.y * S4;                          // first float4 column of this slab
  const int sw = min(S4, n4 - s0);
  const int cl = threadIdx.x % S4, rl = threadIdx.x / S4;
  const int c4 = s0 + cl;
  const bool act = rl < RL && cl < sw;
  // RB rows per thread and trip with all loads ahead of the math; the first trip's rows are requested before `de` is
  // staged (loads unconditional from clamped rows / columns; clamped duplicates are never consumed)
#ifndef GH_DPRE_RB
#define GH_DPRE_RB 4
#endif
  constexpr int RB = GH_DPRE_RB;
  const int c4c = min(c4, n4 - 1);
  const float4* tb = reinterpret_cast<const float4*>(t + (size_t)row0 * Ha) + c4c;
  float4 nxt[RB];
  if (L > 0) {
#pragma unroll
    for (int u = 0; u < RB; ++u) nxt[u] = tb[(size_t)min(rl + u * RL, L - 1) * n4];
  }
  if (dw_in) {
    // second half of the softmax backward, per pair (att_rows_bwd_kernel left the raw dw): de = w (dw - sum_l w dw)
    float* wl = des + (size_t)Lmax * C;                    // [L][C] softmax weights
    for (int i = threadIdx.x; i < L * C; i += blockDim.x) {
      float v = dw_in[(size_t)row0 * C + i];
      for (int rg = 1; rg < dw_ranges; ++rg) v += dw_in[(size_t)rg * (size_t)dw_range_stride + (size_t)row0 * C + i];
      des[i] = v; wl[i] = wts[(size_t)row0 * C + i];
    }
    __syncthreads();
    float* sums = wl + (size_t)Lmax * C;                   // [C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    for (int c = wave; c < C; c += nwv) {
      float sm = 0.f;
      for (int l = lane; l < L; l += 64) sm += wl[l * C + c] * des[l * C + c];
      for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
      if (lane == 0) sums[c] = sm;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * C; i += blockDim.x) {
      const float v = wl[i] * (des[i] - sums[i % C]);
      des[i] = v;
      if (de_out && blockIdx.y == 0) de_out[(size_t)row0 * C + i] = v;
    }
  } else {
    for (int i = threadIdx.x; i < L * C; i += blockDim.x) des[i] = de[(size_t)row0 * C + i];
  }
  float4 wc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    wc[c] = reinterpret_cast<const float4*>(w2 + (size_t)min(c, C - 1) * Ha)[c4c];
    if (c >= C || !act) wc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 dw[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) dw[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (act) {
    for (int l0 = rl; l0 < L; l0 += RB * RL) {
      float4 cur[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) cur[u] = nxt[u];
      if (l0 + RB * RL < L) {
#pragma unroll
        for (int u = 0; u < RB; ++u) nxt[u] = tb[(size_t)min(l0 + (RB + u) * RL, L - 1) * n4];
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int l = l0 + u * RL;
        if (l >= L) break;
        const size_t m = (size_t)row0 + l;
        const float4 tv = cur[u];
        float4 dt = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < C) {
            const float e = des[l * C + c];
            dt.x += e * wc[c].x; dt.y += e * wc[c].y; dt.z += e * wc[c].z; dt.w += e * wc[c].w;
            dw[c].x += e * tv.x; dw[c].y += e * tv.y; dw[c].z += e * tv.z; dw[c].w += e * tv.w;
          }
        const float4 dp = make_float4(dt.x * (1.f - tv.x * tv.x), dt.y * (1.f - tv.y * tv.y), dt.z * (1.f - tv.z * tv.z),
                                      dt.w * (1.f - tv.w * tv.w));
        if (dpre16) {
          typedef float f32x2_t __attribute__((ext_vector_type(2)));
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          const f32x2_t lo = {dp.x, dp.y}, hi = {dp.z, dp.w};
          reinterpret_cast<uint2*>(dpre16 + m * Ha)[c4] = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_t)),
                                                                     __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_t)));
        } else {
          reinterpret_cast<float4*>(dpre + m * Ha)[c4] = dp;
        }
        acc.x += dp.x; acc.y += dp.y; acc.z += dp.z; acc.w += dp.w;
      }
    }
  }
  if (rl < RL) {
    red[(rl * (1 + C) + 0) * S4 + cl] = acc;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < C) red[(rl * (1 + C) + 1 + c) * S4 + cl] = dw[c];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (1 + C) * sw; i += blockDim.x) {
    const int which = i / sw, cc = i % sw;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < RL; ++k) {
      const float4 x = red[(k * (1 + C) + which) * S4 + cc];
      v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
    }
    if (which == 0) reinterpret_cast<float4*>(du + (size_t)b * Ha)[s0 + cc] = v;
    else if (dw2_part) reinterpret_cast<float4*>(dw2_part + ((size_t)b * C + (which - 1)) * Ha)[s0 + cc] = v;
  }
}

int launch_att_dpre(const float* de, const float* w2, const float* t, const int32_t* goff, int m_real, int b, int l,
                    int ha, int heads, float* dpre, float* du, float* dw2_part, hipStream_t s, const float* dw_in,
                    const float* weights, float* de_out, void* dpre16, long long dw_range_stride, int dw_ranges) {
  GH_REQUIRE(ha % 4 == 0 && ha / 4 <= 256, "att_dpre: attention hidden %d must be a multiple of 4 and <= 1024", ha);
  const int n4 = ha / 4;
  // column slabs of <= 32 float4 (512 B of a row per lane group: whole 128-byte lines), ~3 slabs at ha = 300; few pairs
  // (evidence level) keep one slab per 64 columns
  // whole rows per workgroup up to 512 floats: a column slab makes every row a short run (400 B at three slabs) that
  // straddles 128-byte lines shared with the neighbouring slab's workgroup -- measured 3.1 / 3.6 / 3.9 / 4.3 TB/s at
  // 5 / 3 / 2 / 1 slabs (GH_DPRE_SLABS) once the row batches are prefetched
  static int nsl_env = -1;
  if (nsl_env < 0) nsl_env = measure_env("GH_DPRE_SLABS", 0);
  // (round 5: 3 / 5 / 8 slabs for the few-pair launches -- evidence level, 32 workgroups of 30 rows -- measured: no effect on the step)
  const int nsl = nsl_env > 0 ? nsl_env : (n4 + 127) / 128;
  const int S4 = (n4 + nsl - 1) / nsl;
  const int RL = (256 / S4) > 0 ? (256 / S4) : 1;
  const int threads = ((S4 * RL + 63) / 64) * 64;
  GH_REQUIRE(!dw_in || weights, "att_dpre: dw_in needs the softmax weights");
  const size_t lds = (size_t)RL * (1 + heads) * S4 * 16 + (size_t)l * heads * 4 * (dw_in ? 2 : 1) + (dw_in ? 64 : 0);
  prof_begin(s, b < PROF_FEW_GROUPS ? PROF_FEW_ROWS : PROF_ATT_DPRE);
  hipLaunchKernelGGL(att_dpre_kernel, dim3(b, nsl), dim3(threads), lds, s, de, w2, t, goff, l, ha, heads, RL, S4, dpre, du, dw2_part,
                     dw_in, weights, de_out, (unsigned short*)dpre16, dw_range_stride, dw_ranges);
  const double rows = goff ? (double)m_real : (double)b * l;
  prof_end(b < PROF_FEW_GROUPS ? PROF_FEW_ROWS : PROF_ATT_DPRE, 4.0 * (2.0 * rows * ha + rows * heads + (double)b * ha), s);
  GH_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------- ragged helpers (basic_fc_model.py:80-121)
__global__ void __launch_bounds__(256)
seg_offsets_kernel(const int64_t* __restrict__ counts, int B, int32_t* __restrict__ offsets,
                   int32_t* __restrict__ pair2claim, int b1, float* __restrict__ has) {
  __shared__ int total;
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int b = 0; b < B; ++b) { offsets[b] = acc; acc += (int)counts[b]; }
    offsets[B] = acc;
    total = acc;
  }
  __syncthreads();
  __threadfence_block();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const int lo = min(offsets[b], b1), hi = min(offsets[b + 1], b1);
    for (int p = lo; p < hi; ++p) pair2claim[p] = b;
  }
  // counts that sum to less than b1 (a caller bug the reference reports as a shape error): the rows beyond the last
  // claim map to claim 0 instead of staying uninitialised -- no out-of-bounds gather downstream
  for (int p = total + threadIdx.x; p < b1; p += blockDim.x) pair2claim[p] = 0;
  if (has)
    for (int b = threadIdx.x; b < B; b += blockDim.x) has[b] = counts[b] > 0 ? 1.f : 0.f;
}

__global__ void __launch_bounds__(256)
seg_broadcast_kernel(const float* __restrict__ src, const int32_t* __restrict__ pair2claim, float* __restrict__ dst,
                     int b1, int X) {
  const int p = blockIdx.x;
  const float* s = src + (size_t)pair2claim[p] * X;
  for (int i = threadIdx.x; i < X; i += blockDim.x) dst[(size_t)p * X + i] = s[i];
}

__global__ void __launch_bounds__(256)
seg_sum_kernel(const float* __restrict__ src, const int32_t* __restrict__ offsets, float* __restrict__ dst, int X) {
  const int b = blockIdx.x;
  const int lo = offsets[b], hi = offsets[b + 1];
  // (few workgroups, a serial walk over <= n_max rows: eight row loads in flight per trip, added in row order)
  for (int i = threadIdx.x; i < X; i += blockDim.x) {
    float acc = 0.f;
    for (int p = lo; p < hi; p += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)min(p + u, hi - 1) * X + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) if (p + u < hi) acc += v[u];
    }
    dst[(size_t)b * X + i] = acc;
  }
}

// dst[b][slot][0:X] = src[offsets[b]+slot] for slot < count(b), else 0   (dst row pitch dst_ld >= X)
__global__ void __launch_bounds__(256)
seg_pad_kernel(const float* __restrict__ src, const int32_t* __restrict__ offsets, float* __restrict__ dst, int n_max,
               int X, int dst_ld) {
  const int b = blockIdx.x / n_max, slot = blockIdx.x % n_max;
  const int lo = offsets[b], cnt = min(offsets[b + 1] - lo, n_max);
  float* d = dst + ((size_t)b * n_max + slot) * dst_ld;
  if (slot < cnt) {
    const float* s = src + (size_t)(lo + slot) * X;
    for (int i = threadIdx.x; i < X; i += blockDim.x) d[i] = s[i];
  } else {
    for (int i = threadIdx.x; i < X; i += blockDim.x) d[i] = 0.f;
  }
}
__global__ void __launch_bounds__(256)
seg_unpad_kernel(const float* __restrict__ src, const int32_t* __restrict__ offsets, float* __restrict__ dst,
                 int n_max, int X, int src_ld) {
  const int b = blockIdx.x / n_max, slot = blockIdx.x % n_max;
  const int lo = offsets[b], cnt = min(offsets[b + 1] - lo, n_max);
  if (slot >= cnt) return;
  const float* s = src + ((size_t)b * n_max + slot) * src_ld;
  float* d = dst + (size_t)(lo + slot) * X;
  for (int i = threadIdx.x; i < X; i += blockDim.x) d[i] = s[i];
}

// claim vector: sum_l hid[b][l][:] [ids[b][l] > 0] / len[b]   (graph_based_semantic_structure.py:145,153)
__global__ void __launch_bounds__(256)
masked_mean_fwd_kernel(const float* __restrict__ hid, const int32_t* __restrict__ ids, const float* __restrict__ lens,
                       float* __restrict__ dst, int L, int H) {
  const int b = blockIdx.x;
  const float inv = 1.f / lens[b];
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    float acc = 0.f;
    for (int l = 0; l < L; l += 8) {
      float v[8];
      int id[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int lc = min(l + u, L - 1);
        id[u] = ids[b * L + lc];
        v[u] = hid[((size_t)b * L + lc) * H + i];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) if (l + u < L && id[u] > 0) acc += v[u];
    }
    dst[(size_t)b * H + i] = acc * inv;
  }
}
__global__ void __launch_bounds__(256)
masked_mean_bwd_kernel(const float* __restrict__ g, const int32_t* __restrict__ ids, const float* __restrict__ lens,
                       float* __restrict__ dhid, int L, int H) {
  const int b = blockIdx.x / L, l = blockIdx.x % L;
  const float sc = (ids[b * L + l] > 0) ? 1.f / lens[b] : 0.f;
  for (int i = threadIdx.x; i < H; i += blockDim.x) dhid[((size_t)b * L + l) * H + i] = g[(size_t)b * H + i] * sc;
}

// ---------------------------------------------------------------------------- evidence-level assembly
// graph_based_semantic_structure.py:157-170,195-215 in one launch: right[b][slot] = [ pad_right(avg)[b][slot] |
// article_source_embs(max(src, 0)) ], mask[b][slot] = (sum_r document[b][slot][r] >= 1).  Replaces seg_pad +
// masked_fill + embedding + cat + sum/compare/cast (8 launches).  One workgroup per (claim, slot).
template <typename TS, typename TD>
__global__ void __launch_bounds__(256)
evd_assemble_fwd_kernel(const float* __restrict__ avg, const int32_t* __restrict__ offsets, const float* __restrict__ table,
                        const TS* __restrict__ sources, const TD* __restrict__ document, float* __restrict__ right,
                        float* __restrict__ mask, int n_max, int Xa, int Ds, int R, int table_rows, unsigned int* __restrict__ clamped) {
  const int b = blockIdx.x / n_max, slot = blockIdx.x % n_max;
  const int lo = offsets[b], cnt = min(offsets[b + 1] - lo, n_max);
  float* d = right + (size_t)blockIdx.x * (Xa + Ds);
  if (slot < cnt) {
    const float* sp = avg + (size_t)(lo + slot) * Xa;
    for (int i = threadIdx.x; i < Xa; i += blockDim.x) d[i] = sp[i];
  } else {
    for (int i = threadIdx.x; i < Xa; i += blockDim.x) d[i] = 0.f;
  }
  if (Ds > 0) {
    long long sid = (long long)sources[blockIdx.x];
    // -1 padding -> row 0 (:166-168).  Any other id outside the table would raise in nn.Embedding (:169); here it is clamped
    // for memory safety and counted (gh_clamp_events)
    if ((sid < -1 || sid >= table_rows) && threadIdx.x == 0) atomicAdd(clamped + 2, 1u);
    if (sid < 0) sid = 0;
    if (sid >= table_rows) sid = table_rows - 1;
    const float* tp = table + (size_t)sid * Ds;
    for (int i = threadIdx.x; i < Ds; i += blockDim.x) d[Xa + i] = tp[i];
  }
  if (threadIdx.x < 64) {                                        // slot mask: any node id >= 1 (ids are non-negative)
    long long acc = 0;
    const TD* dp = document + (size_t)blockIdx.x * R;
    for (int r = threadIdx.x; r < R; r += 64) acc += (long long)dp[r];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) mask[blockIdx.x] = acc >= 1 ? 1.f : 0.f;
  }
}

// backward: d_avg = unpad(g[:, :, :Xa]);  d_table[s] += sum over the slots with source s of g[:, :, Xa:], summed in
// slot order by the FIRST slot that holds s (deterministic: no float atomics, no sort)
template <typename TS>
__global__ void __launch_bounds__(256)
evd_assemble_bwd_kernel(const float* __restrict__ g, const int32_t* __restrict__ offsets, const TS* __restrict__ sources,
                        float* __restrict__ d_avg, float* __restrict__ d_table, int n_slots, int n_max, int Xa, int Ds, int table_rows) {
  extern __shared__ __attribute__((aligned(16))) int sids[];
  const int me = blockIdx.x;
  const int b = me / n_max, slot = me % n_max;
  const int lo = offsets[b], cnt = min(offsets[b + 1] - lo, n_max);
  const float* gp = g + (size_t)me * (Xa + Ds);
  if (slot < cnt && d_avg) {
    float* d = d_avg + (size_t)(lo + slot) * Xa;
    for (int i = threadIdx.x; i < Xa; i += blockDim.x) d[i] = gp[i];
  }
  // pairs beyond the claim's n_max-th evidence have no slot: their gradient rows are zeros, written by the claim's last slot (the
  // caller's d_avg needs no fill in front of this kernel)
  if (slot == n_max - 1 && d_avg) {
    const size_t z0 = (size_t)(lo + n_max) * Xa, z1 = (size_t)offsets[b + 1] * Xa;
    for (size_t i = z0 + threadIdx.x; i < z1; i += blockDim.x) d_avg[i] = 0.f;
  }
  if (Ds <= 0 || !d_table) return;
  // Padding slots (slot >= the claim's evidence count) are masked out of the evidence-level softmax, so their gradient
  // rows are exact zeros: they neither own nor join a source row (-1 never matches).  Without this every padding slot
  // maps to source 0 and ONE workgroup walks hundreds of zero rows (177 us at the realistic evidence counts).
  for (int i = threadIdx.x; i < n_slots; i += blockDim.x) {
    const int bi = i / n_max;
    const bool real = (i - bi * n_max) < min(offsets[bi + 1] - offsets[bi], n_max);
    const long long v = (long long)sources[i];
    sids[i] = real ? (v < 0 ? 0 : (v >= table_rows ? table_rows - 1 : (int)v)) : -1;      // same clamp as the forward
  }
  __syncthreads();
  const int mine = sids[me];
  if (mine < 0) return;
  __shared__ int earlier;
  if (threadIdx.x == 0) earlier = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < me; i += blockDim.x)
    if (sids[i] == mine) earlier = 1;
  __syncthreads();
  if (earlier) return;                                            // an earlier slot owns this source's row
  // later slots with the same source as a bit mask (ballot per 64 slots), then summed in slot order
  const int nw = (n_slots + 63) / 64;
  unsigned long long* match = reinterpret_cast<unsigned long long*>(sids + ((n_slots + 1) & ~1));
  for (int w0 = (threadIdx.x >> 6); w0 < nw; w0 += (blockDim.x >> 6)) {
    const int j = w0 * 64 + (threadIdx.x & 63);
    const unsigned long long m = __ballot(j < n_slots && j >= me && sids[j] == mine);
    if ((threadIdx.x & 63) == 0) match[w0] = m;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Ds; i += blockDim.x) {
    float acc = 0.f;
    for (int w0 = me >> 6; w0 < nw; ++w0) {
      unsigned long long m = match[w0];
      while (m) {
        const int j = (w0 << 6) + __builtin_ctzll(m);
        m &= m - 1;
        acc += g[(size_t)j * (Xa + Ds) + Xa + i];
      }
    }
    d_table[(size_t)mine * Ds + i] += acc;
  }
}

// ---------------------------------------------------------------------------- linear layers with a handful of outputs
// y[m][n] = x[m][:] . w[n][:] + b[n]  for n <= 8 (the 2-class head, graph_based_semantic_structure.py:72): one wave per row
__global__ void __launch_bounds__(256)
tiny_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                       float* __restrict__ y, int m, int k, int n) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m) return;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  for (int i = lane; i < k; i += 64) {
    const float xv = x[(size_t)row * k + i];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < n) acc[c] += xv * w[(size_t)c * k + i];
  }
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (c < n) {
      float v = acc[c];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) y[(size_t)row * n + c] = v + (bias ? bias[c] : 0.f);
    }
}
// dx[m][k] = g[m][:] . w[:][k] ;  dw[n][k] += sum_m g[m][n] x[m][k] ;  db[n] += sum_m g[m][n]   (w given as stored, [n][k])
__global__ void __launch_bounds__(256)
tiny_linear_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ g,
                       float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int m, int k, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;            // column k index
  if (i < k) {
    float wc[8], dwc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { wc[c] = c < n ? w[(size_t)c * k + i] : 0.f; dwc[c] = 0.f; }
    for (int r0 = 0; r0 < m; r0 += 8) {
      float xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = x[(size_t)min(r0 + u, m - 1) * k + i];       // eight rows in flight
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + u;
        if (r >= m) break;
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < n) { const float gv = g[(size_t)r * n + c]; d += gv * wc[c]; dwc[c] += gv * xv[u]; }
        if (dx) dx[(size_t)r * k + i] = d;
      }
    }
    if (dw)
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < n) dw[(size_t)c * k + i] += dwc[c];
  }
  if (db && blockIdx.x == 0 && threadIdx.x < n) {
    float acc = 0.f;
    for (int r = 0; r < m; ++r) acc += g[(size_t)r * n + threadIdx.x];
    db[threadIdx.x] += acc;
  }
}

// ---------------------------------------------------------------------------- head backward in ONE launch (composite path)
// graph_based_semantic_structure.py:69-74: phi = out1(out0([left | att])) with no activation between.  Backward of both layers' inputs:
//   d_y0[b][k]   = sum_c g_phi[b][c] w1[c][k]                       (H values per claim; tiny: every workgroup makes its own copy in LDS)
//   d_in[b][j]   = sum_k d_y0[b][k] w0[k][j]                        (j < E = Xl + X1: the first Xl go to dx0, the rest to dx1)
//   dw1[c][k]   += sum_b g_phi[b][c] y0[b][k],  db1[c] += sum_b g_phi[b][c]
// It replaces tiny_linear_bwd (5 workgroups walking the claims serially), a 48-workgroup split-K GEMM and its finish kernel --
// 33 us of launch latency for 0.07 GFLOP.  One workgroup per 64 columns j: wave kg takes the k range [kg H/4, (kg+1) H/4), a lane one
// column, 32 claims at a time in registers; w0 is read once (coalesced along j), d_y0 comes from LDS as wave-uniform 16-byte reads.
// The weight gradients of out0 (d_y0^T [left | att]) stay with the side stream's TN launch, which reads the d_y0 this kernel writes.
// CPT = 4 (round 6): a lane owns FOUR consecutive columns (16-byte loads of w0: 256 contiguous bytes per k row and workgroup instead of
// 64).  At h = 768 the head's input is 13 K wide: 832 workgroups of 16 columns each fetched w0 (41 MB) in 64-byte pieces with two
// dependent round trips per thread -- 135 us on the critical path of the configs[4] step; 208 workgroups of 64 columns stream it.
// Used when the width gives >= 192 such workgroups; narrower heads (h = 300: 3556 columns) keep one column per lane.
template <int CPT>
__global__ void __launch_bounds__(256)
head_bwd_kernel(const float* __restrict__ g_phi, const float* __restrict__ y0, const float* __restrict__ w1, const float* __restrict__ w0,
                int B, int H, int C, int Xl, int E, float* __restrict__ d_y0, float* __restrict__ dw1, float* __restrict__ db1,
                float* __restrict__ dx0, int dx0_accumulate, float* __restrict__ dx1, int dy_floats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  float* dy = reinterpret_cast<float*>(dsm);                 // [32][H]; re-used as the partials [16][32][17] after the K loop
  float* w1s = dy + dy_floats;                               // [C][H]
  float* gs = w1s + (size_t)C * H;                           // [B][C]
  float* ys = gs + (size_t)B * C;                            // [nk][B]: y0 columns of this workgroup's share of dw1
  const int tid = threadIdx.x, col = tid & 15, kg = tid >> 4;       // 16 column groups x 16 k ranges per workgroup
  const int j = (blockIdx.x * 16 + col) * CPT;
  const int jc = min(j, E - CPT);
  const int H4 = H / 4;
  const int kq0 = (kg * H4) / 16, kq1 = ((kg + 1) * H4) / 16;     // this thread's range of float4 k-groups
  // everything small is staged once with independent loads (a per-element walk over g_phi / w1 / y0 in global memory put one
  // memory latency into every iteration: 37 us for this kernel)
  const int G = gridDim.x;
  const int nk = dw1 ? max(0, (H - (int)blockIdx.x + G - 1) / G) : 0;      // workgroup w owns the k with k % G == w
  for (int i = tid; i < C * H; i += 256) w1s[i] = w1[i];
  for (int i = tid; i < B * C; i += 256) gs[i] = g_phi[i];
  for (int i = tid; i < nk * B; i += 256) { const int kk = i / B, b = i - kk * B; ys[i] = y0[(size_t)b * H + blockIdx.x + kk * G]; }
  __syncthreads();
  for (int i = tid; i < nk * C; i += 256) {
    const int kk = i / C, c = i - kk * C;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += gs[b * C + c] * ys[kk * B + b];
    dw1[(size_t)c * H + blockIdx.x + kk * G] += acc;
  }
  if (db1 && blockIdx.x == 0 && tid < C) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += gs[b * C + tid];
    db1[tid] += acc;
  }
  for (int b0 = 0; b0 < B; b0 += 32) {
    const int nb = min(32, B - b0);
    __syncthreads();                                         // (the previous chunk's partials are consumed)
    for (int i = tid; i < 32 * H; i += 256) {
      const int b = i / H, k = i - b * H;
      float v = 0.f;
      if (b < nb)
        for (int c = 0; c < C; ++c) v += gs[(b0 + b) * C + c] * w1s[c * H + k];
      dy[i] = v;
      if (blockIdx.x == 0 && b < nb) d_y0[(size_t)(b0 + b) * H + k] = v;
    }
    __syncthreads();
    float acc[32][CPT];
#pragma unroll
    for (int b = 0; b < 32; ++b)
#pragma unroll
      for (int c = 0; c < CPT; ++c) acc[b][c] = 0.f;
    // several k-groups (4 rows of w0 each) requested before any of them is used: one memory round trip per thread for H <= 512
    constexpr int U = CPT == 1 ? 8 : 4;
    for (int kq = kq0; kq < kq1; kq += U) {
      float wv[U][4][CPT];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = 4 * min(kq + u, kq1 - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (CPT == 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(w0 + (size_t)(k + i) * E + jc);
            wv[u][i][0] = t4.x; wv[u][i][1] = t4.y; wv[u][i][2] = t4.z; wv[u][i][3] = t4.w;
          } else {
            wv[u][i][0] = w0[(size_t)(k + i) * E + jc];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (kq + u >= kq1) break;
        const int k = 4 * (kq + u);
#pragma unroll
        for (int b = 0; b < 32; ++b) {
          const float4 d4 = *reinterpret_cast<const float4*>(dy + b * H + k);
#pragma unroll
          for (int c = 0; c < CPT; ++c)
            acc[b][c] += d4.x * wv[u][0][c] + d4.y * wv[u][1][c] + d4.z * wv[u][2][c] + d4.w * wv[u][3][c];
        }
      }
    }
    __syncthreads();                                         // every wave is done with dy: it becomes the partial buffer
    float* red = dy;                                         // [16 k ranges][32][16 column groups + 1], one of a lane's CPT columns at a time
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      if (c > 0) __syncthreads();
#pragma unroll
      for (int b = 0; b < 32; ++b) red[(kg * 32 + b) * 17 + col] = acc[b][c];
      __syncthreads();
      for (int i = tid; i < 32 * 16; i += 256) {
        const int b = i >> 4, l = i & 15, jj = (blockIdx.x * 16 + l) * CPT + c;
        if (b >= nb || jj >= E) continue;
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v += red[(r * 32 + b) * 17 + l];
        if (jj < Xl) {
          if (dx0) { float* o = dx0 + (size_t)(b0 + b) * Xl + jj; *o = dx0_accumulate ? *o + v : v; }
        } else if (dx1) {
          dx1[(size_t)(b0 + b) * (E - Xl) + (jj - Xl)] = v;
        }
      }
    }
  }
}

// LDS bytes of head_bwd_kernel for these sizes (0: does not fit -- the caller keeps the three-launch path)
static bool head_bwd_wide(int E, const float* w0) {      // four columns per lane: 16-byte rows and enough workgroups to fill the chip
  return E % 4 == 0 && (E + 63) / 64 >= 192 && (reinterpret_cast<uintptr_t>(w0) & 15) == 0;
}
size_t head_bwd_lds(int B, int H, int C, int E, const float* w0) {
  const int G = head_bwd_wide(E, w0) ? (E + 63) / 64 : (E + 15) / 16;
  const size_t dyf = (size_t)32 * H > (size_t)16 * 32 * 17 ? (size_t)32 * H : (size_t)16 * 32 * 17;
  const size_t fl = dyf + (size_t)C * H + (size_t)B * C + (size_t)((H + G - 1) / G) * B;
  return fl * 4 <= 160 * 1024 ? fl * 4 : 0;
}

int launch_head_bwd(const float* g_phi, const float* y0, const float* w1, const float* w0, int B, int H, int C, int Xl, int E,
                    float* d_y0, float* dw1, float* db1, float* dx0, int dx0_accumulate, float* dx1, hipStream_t s) {
  GH_REQUIRE(B > 0 && H > 0 && H % 4 == 0, "head_bwd: hidden width %d must be a multiple of 4", H);
  GH_REQUIRE(C >= 1 && C <= 8 && Xl >= 0 && Xl <= E && d_y0, "head_bwd: bad sizes (C=%d Xl=%d E=%d)", C, Xl, E);
  const size_t lds = head_bwd_lds(B, H, C, E, w0);
  GH_REQUIRE(lds > 0, "head_bwd: B=%d, H=%d do not fit the kernel's LDS staging", B, H);
  if (lds > 64 * 1024) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)head_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)head_bwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
  }
  const int dyf = 32 * H > 16 * 32 * 17 ? 32 * H : 16 * 32 * 17;
  if (head_bwd_wide(E, w0))
    hipLaunchKernelGGL(head_bwd_kernel<4>, dim3((E + 63) / 64), dim3(256), lds, s, g_phi, y0, w1, w0, B, H, C, Xl, E, d_y0, dw1, db1, dx0,
                       dx0_accumulate, dx1, dyf);
  else
    hipLaunchKernelGGL(head_bwd_kernel<1>, dim3((E + 15) / 16), dim3(256), lds, s, g_phi, y0, w1, w0, B, H, C, Xl, E, d_y0, dw1, db1, dx0,
                       dx0_accumulate, dx1, dyf);
  GH_LAUNCH_CHECK();
  return 0;
}

int launch_tiny_linear_fwd(const float* x, const float* w, const float* bias, float* y, int m, int k, int n, hipStream_t s) {
  hipLaunchKernelGGL(tiny_linear_fwd_kernel, dim3((m + 3) / 4), dim3(256), 0, s, x, w, bias, y, m, k, n);
  GH_LAUNCH_CHECK();
  return 0;
}
int launch_tiny_linear_bwd(const float* x, const float* w, const float* g, float* dx, float* dw, float* db, int m, int k, int n,
                           hipStream_t s) {
  hipLaunchKernelGGL(tiny_linear_bwd_kernel, dim3((k + 63) / 64), dim3(64), 0, s, x, w, g, dx, dw, db, m, k, n);
  GH_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------- flat Adam (torch.optim.Adam, L2 weight decay)
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            size_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float w = p[i];
    const float gr = g[i] * gscale + wd * w;
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    // torch: denom = sqrt(v)/sqrt(bias_correction2) + eps ; p -= (lr / bias_correction1) * m / denom
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = w - (lr / bc1) * (mi / denom);
  }
}

// Device counters of clamped out-of-range inputs (include/get_hip.h gh_clamp_events): four unsigned ints per device, allocated
// on first use, handed to the kernels that clamp (cross_entropy: [0], left_assemble: [1], evd_assemble: [2]).
static std::mutex g_clamp_mu;
static std::unordered_map<int, unsigned int*> g_clamp_buf;
unsigned int* clamp_counter() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_clamp_mu);
  auto it = g_clamp_buf.find(dev);
  if (it != g_clamp_buf.end()) return it->second;
  unsigned int* p = nullptr;
  if (hipMalloc((void**)&p, 4 * sizeof(unsigned int)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipMemset(p, 0, 4 * sizeof(unsigned int)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return nullptr; }
  g_clamp_buf.emplace(dev, p);
  return p;
}

}  // namespace gh

using namespace gh;

extern "C" int gh_clamp_events(int64_t* out, int reset) {
  GH_REQUIRE(out, "clamp_events: NULL output");
  unsigned int* p = clamp_counter();
  GH_REQUIRE(p, "clamp_events: cannot allocate the counter");
  unsigned int h[4] = {0, 0, 0, 0};
  GH_CHECK_HIP(hipDeviceSynchronize());
  GH_CHECK_HIP(hipMemcpy(h, p, sizeof(h), hipMemcpyDeviceToHost));
  for (int i = 0; i < 4; ++i) out[i] = (int64_t)h[i];
  if (reset) GH_CHECK_HIP(hipMemset(p, 0, sizeof(h)));
  return 0;
}

extern "C" int gh_abi_version(void) { return GH_ABI_VERSION; }
extern "C" const char* gh_last_error(void) { return g_err; }

extern "C" int gh_profile_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}
// out[PROF_NTAGS][3] = {total ms, total work, launches}; waits for the recorded events, then resets.
extern "C" int gh_profile_select(uint32_t tag_mask) {
  g_prof_mask = tag_mask;
  return 0;
}

extern "C" int gh_profile_collect(double* out, int rows) {
  GH_REQUIRE(rows >= PROF_NTAGS, "profile_collect: need %d rows", (int)PROF_NTAGS);
  for (int i = 0; i < rows * 3; ++i) out[i] = 0.0;
  for (auto& r : g_prof_recs) {
    float ms = 0.f;
    GH_CHECK_HIP(hipEventSynchronize(r.b));
    GH_CHECK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    out[r.tag * 3 + 0] += ms;
    out[r.tag * 3 + 1] += r.work;
    out[r.tag * 3 + 2] += 1.0;
    g_prof_pool.push_back(r.a);
    g_prof_pool.push_back(r.b);
  }
  g_prof_recs.clear();
  return 0;
}

extern "C" int gh_transpose(const float* w, float* wt, int rows, int cols, gh_stream_t stream) {
  GH_REQUIRE(rows > 0 && cols > 0, "transpose: bad sizes %d x %d", rows, cols);
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, w,
                     wt, rows, cols);
  GH_LAUNCH_CHECK();
  return 0;
}

extern "C" int gh_weights_refresh(int n, const void* const* src, void* const* dst, void* const* dst_w16, void* const* dst_t16,
                                  const int* rows, const int* cols, gh_stream_t stream) {
  GH_REQUIRE(n >= 0 && (n == 0 || (src && rows && cols)), "weights_refresh: NULL arrays");
  for (int i0 = 0; i0 < n; i0 += 32) {
    TransposeBatch B;
    B.n = (n - i0 < 32) ? n - i0 : 32;
    int tiles = 0;
    for (int i = 0; i < B.n; ++i) {
      GH_REQUIRE(rows[i0 + i] > 0 && cols[i0 + i] > 0 && src[i0 + i], "weights_refresh: bad matrix at %d", i0 + i);
      B.src[i] = (const float*)src[i0 + i]; B.dst[i] = dst ? (float*)dst[i0 + i] : nullptr;
      B.w16[i] = dst_w16 ? (unsigned short*)dst_w16[i0 + i] : nullptr;
      B.t16[i] = dst_t16 ? (unsigned short*)dst_t16[i0 + i] : nullptr;
      B.rows[i] = rows[i0 + i]; B.cols[i] = cols[i0 + i];
      B.tile0[i] = tiles;
      tiles += ((rows[i0 + i] + 31) / 32) * ((cols[i0 + i] + 31) / 32);
    }
    B.tile0[B.n] = tiles;
    hipLaunchKernelGGL(transpose_batch_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, B);
    GH_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int gh_transpose_batch(int n, const void* const* src, void* const* dst, const int* rows, const int* cols,
                                  gh_stream_t stream) {
  GH_REQUIRE(n == 0 || dst, "transpose_batch: NULL dst");
  return gh_weights_refresh(n, src, dst, nullptr, nullptr, rows, cols, stream);
}

extern "C" int gh_seg_offsets(const int64_t* counts, int b, int32_t* offsets, int32_t* pair2claim, int b1, float* has,
                              gh_stream_t stream) {
  GH_REQUIRE(b > 0, "seg_offsets: b=%d", b);
  hipLaunchKernelGGL(seg_offsets_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, counts, b, offsets, pair2claim, b1, has);
  GH_LAUNCH_CHECK();
  return 0;
}
extern "C" int gh_seg_broadcast(const float* src, const int32_t* pair2claim, float* dst, int b1, int x,
                                gh_stream_t stream) {
  if (b1 <= 0) return 0;
  hipLaunchKernelGGL(seg_broadcast_kernel, dim3(b1), dim3(256), 0, (hipStream_t)stream, src, pair2claim, dst, b1, x);
  GH_LAUNCH_CHECK();
  return 0;
}
extern "C" int gh_seg_sum(const float* src, const int32_t* offsets, float* dst, int b, int x, gh_stream_t stream) {
  if (b <= 0) return 0;
  hipLaunchKernelGGL(seg_sum_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, src, offsets, dst, x);
  GH_LAUNCH_CHECK();
  return 0;
}
extern "C" int gh_seg_pad(const float* src, const int32_t* offsets, float* dst, int b, int n_max, int x, int dst_ld,
                          gh_stream_t stream) {
  if (b <= 0) return 0;
  hipLaunchKernelGGL(seg_pad_kernel, dim3(b * n_max), dim3(256), 0, (hipStream_t)stream, src, offsets, dst, n_max, x, dst_ld);
  GH_LAUNCH_CHECK();
  return 0;
}
extern "C" int gh_seg_unpad(const float* src, const int32_t* offsets, float* dst, int b, int n_max, int x, int src_ld,
                            gh_stream_t stream) {
  if (b <= 0) return 0;
  hipLaunchKernelGGL(seg_unpad_kernel, dim3(b * n_max), dim3(256), 0, (hipStream_t)stream, src, offsets, dst, n_max, x, src_ld);
  GH_LAUNCH_CHECK();
  return 0;
}
extern "C" int gh_masked_mean_fwd(const float* hid, const int32_t* ids, const float* lens, float* dst, int b, int l,
                                  int h, gh_stream_t stream) {
  if (b <= 0) return 0;
  hipLaunchKernelGGL(masked_mean_fwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, hid, ids, lens, dst, l, h);
  GH_LAUNCH_CHECK();
  return 0;
}
extern "C" int gh_masked_mean_bwd(const float* g, const int32_t* ids, const float* lens, float* dhid, int b, int l,
                                  int h, gh_stream_t stream) {
  if (b <= 0) return 0;
  hipLaunchKernelGGL(masked_mean_bwd_kernel, dim3(b * l), dim3(256), 0, (hipStream_t)stream, g, ids, lens, dhid, l, h);
  GH_LAUNCH_CHECK();
  return 0;
}

extern "C" int gh_adam_step(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int step, float grad_scale,
                            gh_stream_t stream) {
  GH_REQUIRE(step >= 1, "adam_step: step=%d must be >= 1", step);
  if (count <= 0) return 0;
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int grid = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
  prof_begin((hipStream_t)stream, PROF_ADAM);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (size_t)count, lr, beta1,
                     beta2, eps, weight_decay, bc1, (float)sqrt(bc2), grad_scale);
  prof_end(PROF_ADAM, 28.0 * (double)count, (hipStream_t)stream);
  GH_LAUNCH_CHECK();
  return 0;
}

extern "C" int gh_evd_assemble_fwd(const float* avg, const int32_t* offsets, const float* table, int table_rows, const void* sources,
                                   int sources_i64, const void* document, int document_i64, int b, int n_max, int xa, int ds,
                                   int r, float* right, float* mask, gh_stream_t stream) {
  GH_REQUIRE(b > 0 && n_max > 0 && xa > 0 && ds >= 0 && r > 0, "evd_assemble_fwd: bad sizes");
  GH_REQUIRE(ds == 0 || (table && sources), "evd_assemble_fwd: article-source width %d needs a table and source ids", ds);
  GH_REQUIRE(ds == 0 || table_rows > 0, "evd_assemble_fwd: table_rows must give the article-source table's row count");
  hipStream_t s = (hipStream_t)stream;
  unsigned int* cl = clamp_counter();
  GH_REQUIRE(cl, "evd_assemble_fwd: cannot allocate the clamp counter");
  const dim3 grid(b * n_max), blk(256);
#define GH_EA(TS, TD) hipLaunchKernelGGL((evd_assemble_fwd_kernel<TS, TD>), grid, blk, 0, s, avg, offsets, table, (const TS*)sources, \
                                         (const TD*)document, right, mask, n_max, xa, ds, r, table_rows, cl)
  if (sources_i64 && document_i64) GH_EA(int64_t, int64_t);
  else if (sources_i64) GH_EA(int64_t, int32_t);
  else if (document_i64) GH_EA(int32_t, int64_t);
  else GH_EA(int32_t, int32_t);
#undef GH_EA
  GH_LAUNCH_CHECK();
  return 0;
}

extern "C" int gh_evd_assemble_bwd(const float* g, const int32_t* offsets, const void* sources, int sources_i64, int table_rows, int b, int n_max,
                                   int xa, int ds, float* d_avg, float* d_table, gh_stream_t stream) {
  GH_REQUIRE(b > 0 && n_max > 0 && xa > 0 && ds >= 0, "evd_assemble_bwd: bad sizes");
  GH_REQUIRE(ds == 0 || !d_table || table_rows > 0, "evd_assemble_bwd: table_rows must give the article-source table's row count");
  const int n_slots = b * n_max;
  // every workgroup stages the source id of every slot (4 B) + a match bit per slot in LDS: 160 KB hold 38 000 slots,
  // i.e. 1266 claims x 30 evidence slots per call
  GH_REQUIRE(n_slots <= 38000, "evd_assemble_bwd: %d claim x evidence slots (max 38000 per call: split the batch)", n_slots);
  hipStream_t s = (hipStream_t)stream;
  const size_t lds_bytes = (size_t)((n_slots + 1) & ~1) * 4 + (size_t)((n_slots + 63) / 64) * 8;
  if (lds_bytes > 48 * 1024) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)evd_assemble_bwd_kernel<int64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)evd_assemble_bwd_kernel<int32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
  }
  if (sources_i64)
    hipLaunchKernelGGL((evd_assemble_bwd_kernel<int64_t>), dim3(n_slots), dim3(256), lds_bytes, s, g, offsets,
                       (const int64_t*)sources, d_avg, d_table, n_slots, n_max, xa, ds, table_rows);
  else
    hipLaunchKernelGGL((evd_assemble_bwd_kernel<int32_t>), dim3(n_slots), dim3(256), lds_bytes, s, g, offsets,
                       (const int32_t*)sources, d_avg, d_table, n_slots, n_max, xa, ds, table_rows);
  GH_LAUNCH_CHECK();
  return 0;
}
