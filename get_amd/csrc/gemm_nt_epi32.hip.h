// Straight-line, software-pipelined epilogue of the fp32-stream configurations of gemm_nt_kernel (MODE != 2, MI == 2: the
// 64 x 320, 64 x 160 and 32 x 320 tiles).  Included INSIDE the kernel body (gemm_nt.hip.h); sets `f32_done`.
//
// The plain passes (`pass` in gemm_nt.hip.h) load a chunk of four items inside `if (row < M ...)` branches, compute, store, and
// start the next chunk: with branches around the memory instructions the compiler cannot count its waits, every chunk drains
// to vmcnt(0) -- and vmcnt counts STORES too, so the third chunk of a pass waits for the write acknowledgements of the second
// (230 `s_waitcnt vmcnt(0)` in the 64 x 320 kernel; six dependent load + store round trips per tile).  Here, as in
// gemm_nt_pp_epi.hip.h: every global access is a raw buffer access whose lanes outside the problem carry the out-of-range
// offset (loads return 0, stores are dropped), the items of both passes form ONE straight-line sequence, the inputs of item
// k + 2 are requested before item k is computed (three register sets of <= 4 x 16 B), and the waits come out counted.
// Semantics are those of `pass`, kind by kind (same arithmetic, same dead-store rules for the padding rows of the node-compact
// layout, nontemporal output stores where `st4` used them).
{
  constexpr int NITF = (ITEMS + NTHR - 1) / NTHR;      // items per thread and pass
  constexpr int TOT = MI * NITF;
#ifndef GH_E32_PD
#define GH_E32_PD 2
#endif
  constexpr int PD = GH_E32_PD, NSET = PD + 1;      // prefetch distance in items, register sets
  typedef unsigned e32_u32x4 __attribute__((ext_vector_type(4)));
  // (the thread id through an opaque move: without it the compiler hoists the items' address arithmetic -- loop invariant as far
  //  as it can see -- above the K loop, where ~60 live values spill inside the MFMA stream)
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  auto f_item = [&](int mi_, int it, int& row, int& col, int& rr) __attribute__((always_inline)) {
    const int i = tid_e + it * NTHR;
    rr = i / C4;
    const int c4 = i - rr * C4;
    col = 4 * c4;
    row = m0 + (rr >> 4) * 16 * MI + mi_ * 16 + (rr & 15);
    return (ITEMS % NTHR == 0 || i < ITEMS) && c4 < N4 && row < M;
  };
  // (whole-vector cast: __builtin_bit_cast(float, u[i]) on an ext-vector ELEMENT reads element 0 for every i with this compiler --
  //  clang 22 / ROCm 7.2 emits `load float, ptr %vec` for each of them)
  auto as_f4 = [](const e32_u32x4 u) __attribute__((always_inline)) {
    const f32x4 f = __builtin_bit_cast(f32x4, u);
    return make_float4(f[0], f[1], f[2], f[3]);
  };
  auto as_u4 = [](const float4 v) __attribute__((always_inline)) {
    return e32_u32x4{__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w)};
  };
  auto fast = [&](auto EPI, auto ACCF, auto GINF) __attribute__((always_inline)) {
    constexpr int E = decltype(EPI)::value;
    constexpr bool ACC = decltype(ACCF)::value, GIN = decltype(GINF)::value, ATT = E == EPI_ATT;
    constexpr int NIN = E == EPI_STORE ? (ACC ? 1 : 0) : (E == EPI_SIGMOID_Z || ATT) ? 0 : E == EPI_SIGMOID_R ? 1 : E == EPI_TANH_H ? 2 : 3;
    constexpr int NOUT = (E == EPI_STORE || E == EPI_SIGMOID_Z || ATT) ? 1 : E == EPI_GATE_PRE ? 3 : 2;
    constexpr int NTAUX = (E == EPI_GATE_PRE || ATT) ? 0 : 2;      // nontemporal output stores, as st4 (the gate head's and t's were plain)
    const void* i0 = E == EPI_STORE ? (const void*)C : (const void*)in0;
    const void* i2 = E == EPI_BWD_DRX ? (const void*)out1 : (const void*)P.in2;
    const rsrc_t rs_i0 = __builtin_amdgcn_make_buffer_rsrc((void*)(NIN >= 1 ? i0 : (const void*)C), 0, 0x7fffffff, 0x00020000);
    const rsrc_t rs_i1 = __builtin_amdgcn_make_buffer_rsrc((void*)(NIN >= 2 ? (const void*)in1 : (const void*)C), 0, 0x7fffffff, 0x00020000);
    const rsrc_t rs_i2 = __builtin_amdgcn_make_buffer_rsrc((void*)(NIN >= 3 ? i2 : (const void*)C), 0, 0x7fffffff, 0x00020000);
    const rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(GIN ? (const void*)P.gin : ATT ? (const void*)P.u : (const void*)C), 0, 0x7fffffff, 0x00020000);
    const rsrc_t rs_o0 = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, 0x7fffffff, 0x00020000);
    const rsrc_t rs_o1 = __builtin_amdgcn_make_buffer_rsrc((void*)(NOUT >= 2 ? (void*)out1 : (void*)C), 0, 0x7fffffff, 0x00020000);
    const rsrc_t rs_o2 = __builtin_amdgcn_make_buffer_rsrc((void*)(NOUT >= 3 ? (void*)P.out2 : (void*)C), 0, 0x7fffffff, 0x00020000);
    e32_u32x4 qa[NSET], qb[NSET], qc[NSET], qg[NSET];
    // EPI_ATT: byte offset of every item's u row (the row's pair / claim through rowg, else row / R) -- looked up for all items
    // up front: a dependent index load inside the item sequence would drain the counter at every item
    unsigned uoff[ATT ? TOT : 1];
    if constexpr (ATT) {
#pragma unroll
      for (int idx = 0; idx < TOT; ++idx) {
        int row, col, rr;
        const bool ok = f_item(idx / NITF, idx % NITF, row, col, rr);
        const int rc_ = min(row, M - 1);
        const int ur = P.rowg ? P.rowg[rc_] : rc_ / P.R;
        uoff[idx] = ok ? (unsigned)(ur * P.ldu + col) * 4u : OOB;
      }
    }
    auto load = [&](auto IDX) __attribute__((always_inline)) {
      constexpr int idx = decltype(IDX)::value, mi_ = idx / NITF, it = idx % NITF, s_ = idx % NSET;
      int row, col, rr;
      const bool ok = f_item(mi_, it, row, col, rr);
      const unsigned vo = ok ? (unsigned)(row * ldc + col) * 4u : OOB;
      if constexpr (NIN >= 1) qa[s_] = __builtin_bit_cast(e32_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_i0, vo, 0, 0));
      if constexpr (NIN >= 2) qb[s_] = __builtin_bit_cast(e32_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_i1, vo, 0, 0));
      if constexpr (NIN >= 3) qc[s_] = __builtin_bit_cast(e32_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_i2, vo, 0, 0));
      if constexpr (GIN) qg[s_] = __builtin_bit_cast(e32_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, vo, 0, 0));
      if constexpr (ATT) qg[s_] = __builtin_bit_cast(e32_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, uoff[idx], 0, 0));
    };
    auto compute = [&](auto IDX) __attribute__((always_inline)) {
      constexpr int idx = decltype(IDX)::value, mi_ = idx / NITF, it = idx % NITF, s_ = idx % NSET;
      int row, col, rr;
      const bool ok = f_item(mi_, it, row, col, rr);
      const unsigned vo = ok ? (unsigned)(row * ldc + col) * 4u : OOB;
      // (items beyond the staged rows read a clamped LDS address; their stores are dropped)
      float* sp = ep + min(rr, 16 * WM - 1) * EP_PITCH + min(col, BN - 4);
      const float4 v4 = *reinterpret_cast<const float4*>(sp);
      const float4 b4 = *reinterpret_cast<const float4*>(bsum + min(col, BN - 4));
      float4 w = make_float4(v4.x + b4.x, v4.y + b4.y, v4.z + b4.z, v4.w + b4.w);
      const bool pad = pad_rows > 0 && row >= pad_rows;
      auto st = [&](const rsrc_t rs, const float4 v, bool live) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b128(as_u4(v), rs, live ? vo : OOB, 0, NTAUX);
      };
      if (E == EPI_STORE) {
        if (drop_mode == 3)
          w = drop4(w, drop_seed, (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col), drop_thresh, drop_scale);
        if constexpr (ACC) { const float4 x = as_f4(qa[s_]); w.x += x.x; w.y += x.y; w.z += x.z; w.w += x.w; }
        st(rs_o0, w, true);
      } else if (E == EPI_SIGMOID_Z) {
        st(rs_o0, make_float4(sigmoidf_(w.x), sigmoidf_(w.y), sigmoidf_(w.z), sigmoidf_(w.w)), true);
      } else if (E == EPI_SIGMOID_R) {
        const float4 x = as_f4(qa[s_]);
        const float4 r4 = make_float4(sigmoidf_(w.x), sigmoidf_(w.y), sigmoidf_(w.z), sigmoidf_(w.w));
        st(rs_o0, r4, !pad);            // padding rows: r is backward-only and the backward never touches a padding row
        st(rs_o1, make_float4(r4.x * x.x, r4.y * x.y, r4.z * x.z, r4.w * x.w), true);
      } else if (E == EPI_TANH_H) {
        const float4 z = as_f4(qa[s_]), x = as_f4(qb[s_]);
        const float4 h = make_float4(tanhf_(w.x), tanhf_(w.y), tanhf_(w.z), tanhf_(w.w));
        float4 y = make_float4(h.x * z.x + x.x * (1.f - z.x), h.y * z.y + x.y * (1.f - z.y),
                               h.z * z.z + x.z * (1.f - z.z), h.w * z.w + x.w * (1.f - z.w));
        st(rs_o0, h, !pad);
        st(rs_o1, y, !(pad && out_dead));
        if (scorer) {    // the word scorer sees dropout(out) (its own input dropout, wrapper.py:189-190)
          if (drop_mode == 2)
            y = drop4(y, drop_seed, (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col), drop_thresh, drop_scale);
          if (ok) *reinterpret_cast<float4*>(sp) = y;
        }
      } else if (E == EPI_BWD_DRX) {
        const float4 x = as_f4(qa[s_]), r4 = as_f4(qb[s_]);
        float4 d = as_f4(qc[s_]);
        st(rs_o0, make_float4(w.x * x.x * r4.x * (1.f - r4.x), w.y * x.y * r4.y * (1.f - r4.y),
                              w.z * x.z * r4.z * (1.f - r4.z), w.w * x.w * r4.w * (1.f - r4.w)), true);
        d.x += w.x * r4.x; d.y += w.y * r4.y; d.z += w.z * r4.z; d.w += w.w * r4.w;
        st(rs_o1, d, true);
      } else if (E == EPI_GATE_PRE) {
        if (drop_mode == 3)      // g is the gradient w.r.t. a dropped-out input (the producing cell's dX)
          w = drop4(w, drop_seed, (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col), drop_thresh, drop_scale);
        if constexpr (GIN) { const float4 a4 = as_f4(qg[s_]); w.x += a4.x; w.y += a4.y; w.z += a4.z; w.w += a4.w; }
        const float4 Z = as_f4(qa[s_]), Hh = as_f4(qb[s_]), X = as_f4(qc[s_]);
        float4 a, b, c;
#define GH_E32_ONE(f)                                  \
        a.f = w.f * Z.f * (1.f - Hh.f * Hh.f);             \
        b.f = w.f * (Hh.f - X.f) * Z.f * (1.f - Z.f);      \
        c.f = w.f * (1.f - Z.f);
        GH_E32_ONE(x) GH_E32_ONE(y) GH_E32_ONE(z) GH_E32_ONE(w)
#undef GH_E32_ONE
        st(rs_o0, a, true); st(rs_o1, b, true); st(rs_o2, c, true);
      } else if (E == EPI_ATT) {
        const float4 u4 = as_f4(qg[s_]);
        const float4 t4 = make_float4(tanhf_(w.x + u4.x), tanhf_(w.y + u4.y), tanhf_(w.z + u4.z), tanhf_(w.w + u4.w));
        st(rs_o0, t4, true);
        if (ok) *reinterpret_cast<float4*>(sp) = t4;
      }
    };
    auto stage = [&](auto MIT) __attribute__((always_inline)) {
      constexpr int mi_ = decltype(MIT)::value;
      if (mi_ > 0) __syncthreads();                  // previous pass consumed (the loop's last barrier covers pass 0)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        *reinterpret_cast<f32x4*>(ep + (wm * 16 + l15) * EP_PITCH + wcol + ni * 16 + 4 * q) = acc[mi_][ni];
      __syncthreads();
    };
    auto step = [&](auto IDX) __attribute__((always_inline)) {
      constexpr int idx = decltype(IDX)::value;
      if constexpr (idx % NITF == 0) stage(std::integral_constant<int, idx / NITF>{});
      if constexpr (idx + PD < TOT) load(std::integral_constant<int, idx + PD>{});
#ifndef GH_E32_NOSB
      __builtin_amdgcn_sched_barrier(0);      // (pin the order of the items)
#endif
      compute(IDX);
#ifndef GH_E32_NOSB
      __builtin_amdgcn_sched_barrier(0);
#endif
      if constexpr (idx % NITF == NITF - 1) { if (rowred) row_reduce(std::integral_constant<int, idx / NITF>{}); }
    };
    load(std::integral_constant<int, 0>{});
    if constexpr (TOT > 1 && PD > 1) load(std::integral_constant<int, 1>{});
    // (TOT <= 24: 64 x 320 tile = 2 x 11, 64 x 160 = 2 x 6, 32 x 320 = 2 x 6)
    static_assert(TOT <= 24, "the unrolled item sequence below");
#define GH_E32_STEP(k) if constexpr (k < TOT) step(std::integral_constant<int, k>{});
    GH_E32_STEP(0) GH_E32_STEP(1) GH_E32_STEP(2) GH_E32_STEP(3) GH_E32_STEP(4) GH_E32_STEP(5) GH_E32_STEP(6) GH_E32_STEP(7)
    GH_E32_STEP(8) GH_E32_STEP(9) GH_E32_STEP(10) GH_E32_STEP(11) GH_E32_STEP(12) GH_E32_STEP(13) GH_E32_STEP(14) GH_E32_STEP(15)
    GH_E32_STEP(16) GH_E32_STEP(17) GH_E32_STEP(18) GH_E32_STEP(19) GH_E32_STEP(20) GH_E32_STEP(21) GH_E32_STEP(22) GH_E32_STEP(23)
#undef GH_E32_STEP
  };
  constexpr std::integral_constant<bool, false> NO{};
  constexpr std::integral_constant<bool, true> YES{};
  if (!(dbg_bits & 128) && io == 0) {
    f32_done = true;
    if (epi == EPI_STORE) { if (accumulate) fast(std::integral_constant<int, EPI_STORE>{}, YES, NO); else fast(std::integral_constant<int, EPI_STORE>{}, NO, NO); }
    else if (epi == EPI_SIGMOID_Z) fast(std::integral_constant<int, EPI_SIGMOID_Z>{}, NO, NO);
    else if (epi == EPI_SIGMOID_R) fast(std::integral_constant<int, EPI_SIGMOID_R>{}, NO, NO);
    else if (epi == EPI_TANH_H) fast(std::integral_constant<int, EPI_TANH_H>{}, NO, NO);
    else if (epi == EPI_BWD_DRX) fast(std::integral_constant<int, EPI_BWD_DRX>{}, NO, NO);
    else if (epi == EPI_GATE_PRE) { if (P.gin) fast(std::integral_constant<int, EPI_GATE_PRE>{}, NO, YES); else fast(std::integral_constant<int, EPI_GATE_PRE>{}, NO, NO); }
    else if (epi == EPI_ATT) fast(std::integral_constant<int, EPI_ATT>{}, NO, NO);
    else f32_done = false;
  }
}
