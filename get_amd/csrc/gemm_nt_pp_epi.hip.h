// Epilogue of gemm_nt_kernel<2, 4, 4, 8, 2> (256 x 256 bf16 tile, 512 threads, ONE workgroup per CU).  (EPI_ATT with other than fp32
// streams leaves `pp_done` false and takes the kernel's plain passes.)
// Included INSIDE the kernel body (gemm_nt.hip.h).  Same whole-row scheme as `pass8` -- eight passes of 32 staged rows, items of
// eight columns, one 16-byte access per bf16 stream and item -- but SOFTWARE-PIPELINED across the passes: with one workgroup per
// CU nothing else runs underneath an epilogue, and a pass that requests its input streams only after its rows are staged waits a
// full memory round trip eight times per tile.  The raw bf16 inputs of pass p + 1 are requested (two register sets of 2 items x
// <= 3 streams x 16 B) before pass p is computed and stored.
//
// Two forms:
//   * pp_fast<kind, accumulate, gin>: every stream of the kind is bf16 (what the bf16 storage pipeline issues).  STRAIGHT-LINE
//     code: every load and store is a raw buffer access whose lanes outside the problem carry the out-of-range offset (loads
//     return 0, stores are dropped), so a pass has no branch around a memory instruction and the waits are COUNTED: vmcnt counts
//     stores too, and a pass that had to drain to vmcnt(0) to see its prefetched inputs also waited for the write
//     acknowledgements of the pass before it (5.4 us per pass measured; 872 `s_waitcnt vmcnt(0)` in the generic form's code).
//   * the generic form (pp_all): run-time stream formats, branches around the accesses -- fp32 streams, c32 copies.
{
  typedef unsigned short pp_u16;
  typedef unsigned pp_u32x4 __attribute__((ext_vector_type(4)));
  const int pp_io = (MODE == 2) ? P.io : 0;
  struct PF8 { float4 a, b; };
  auto pp_cvt = [](const pp_u32x4 u) __attribute__((always_inline)) {
    PF8 r;
    r.a = make_float4(__builtin_bit_cast(float, u[0] << 16), __builtin_bit_cast(float, u[0] & 0xffff0000u),
                      __builtin_bit_cast(float, u[1] << 16), __builtin_bit_cast(float, u[1] & 0xffff0000u));
    r.b = make_float4(__builtin_bit_cast(float, u[2] << 16), __builtin_bit_cast(float, u[2] & 0xffff0000u),
                      __builtin_bit_cast(float, u[3] << 16), __builtin_bit_cast(float, u[3] & 0xffff0000u));
    return r;
  };
  auto pp_pack = [](const float4 a, const float4 b) __attribute__((always_inline)) {
    return pp_u32x4{nt_pack_bf16(a.x, a.y), nt_pack_bf16(a.z, a.w), nt_pack_bf16(b.x, b.y), nt_pack_bf16(b.z, b.w)};
  };
  auto sig4 = [](const float4 v) __attribute__((always_inline)) { return make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)); };
  auto tanh4 = [](const float4 v) __attribute__((always_inline)) { return make_float4(tanhf_(v.x), tanhf_(v.y), tanhf_(v.z), tanhf_(v.w)); };
  auto mul4 = [](const float4 a, const float4 b) __attribute__((always_inline)) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); };
  auto add4 = [](const float4 a, const float4 b) __attribute__((always_inline)) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
  auto mix4 = [](const float4 h, const float4 z, const float4 x) __attribute__((always_inline)) {      // h z + x (1 - z)
    return make_float4(h.x * z.x + x.x * (1.f - z.x), h.y * z.y + x.y * (1.f - z.y), h.z * z.z + x.z * (1.f - z.z), h.w * z.w + x.w * (1.f - z.w));
  };
  auto drx4 = [](const float4 w, const float4 x, const float4 r) __attribute__((always_inline)) {      // w x r (1 - r)
    return make_float4(w.x * x.x * r.x * (1.f - r.x), w.y * x.y * r.y * (1.f - r.y), w.z * x.z * r.z * (1.f - r.z), w.w * x.w * r.w * (1.f - r.w));
  };
  auto pp_head = [](const float4 g, const float4 Z, const float4 Hh, const float4 X, float4& a, float4& b, float4& c) __attribute__((always_inline)) {
#define GH_PP_ONE(f)                                  \
    a.f = g.f * Z.f * (1.f - Hh.f * Hh.f);            \
    b.f = g.f * (Hh.f - X.f) * Z.f * (1.f - Z.f);     \
    c.f = g.f * (1.f - Z.f);
    GH_PP_ONE(x) GH_PP_ONE(y) GH_PP_ONE(z) GH_PP_ONE(w)
#undef GH_PP_ONE
  };
  // item j of pass p: staged row rr = wm' * 16 + l15'  <->  tile row p of wave row wm'; eight columns from col
  auto pp_item = [&](int p_, int j, int& row, int& col, int& rr) __attribute__((always_inline)) {
    const int i = tid + j * NTHR;
    rr = i >> 5;
    col = 8 * (i & 31);
    row = m0 + (rr >> 4) * 16 * MI + p_ * 16 + (rr & 15);
    return row < M && col < N;
  };
  auto pp_stage = [&](auto PT) __attribute__((always_inline)) {
    constexpr int p_ = decltype(PT)::value;
    if (p_ > 0) __syncthreads();                  // previous pass consumed (the K loop's last barrier covers pass 0)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      *reinterpret_cast<f32x4*>(ep + (wm * 16 + l15) * EP_PITCH + wcol + ni * 16 + 4 * q) = acc[p_][ni];
    __syncthreads();
  };

  // ------------------------------------------------------------------ straight-line form: every stream bf16
  auto pp_fast = [&](auto EPI, auto ACCF, auto GINF, auto SCF) __attribute__((always_inline)) {
    constexpr int E = decltype(EPI)::value;
    constexpr bool ACC = decltype(ACCF)::value, GIN = decltype(GINF)::value;
    // SC (EPI_TANH_H with the GSL scorer's projection, one partial per column block): the block's share of the dot product
    // out[row] . w2 is reduced IN REGISTERS -- a thread's eight columns against its eight scorer weights (the same eight for both of
    // its items and every pass: col = 8 (tid & 31)), then a butterfly over the 32 lanes that hold the row -- instead of writing the
    // finished rows back to LDS and reducing them on the MFMA between two more barriers per pass (row_reduce); the store of the
    // partial is one more out-of-range-predicated buffer store, so the passes stay straight-line.
    constexpr bool SC = decltype(SCF)::value;
    constexpr int NIN = E == EPI_STORE ? (ACC ? 1 : 0) : (E == EPI_SIGMOID_Z || E == EPI_ATT) ? 0 : E == EPI_SIGMOID_R ? 1 : E == EPI_TANH_H ? 2 : 3;
    constexpr bool ATT = E == EPI_ATT;      // fp32 streams: u[pair of the row][col] in (two 16-byte halves per item), t out
    constexpr int NOUT = (E == EPI_STORE || E == EPI_SIGMOID_Z) ? 1 : E == EPI_GATE_PRE ? 3 : 2;
    // input streams: STORE+accumulate reads C; BWD_DRX's third input is out1 (dxp += .); GATE_PRE's is in2
    const void* i0 = E == EPI_STORE ? (const void*)C : (const void*)in0;
    const void* i1 = (const void*)in1;
    const void* i2 = E == EPI_BWD_DRX ? (const void*)out1 : (const void*)P.in2;
    const rsrc_t rs_i0 = __builtin_amdgcn_make_buffer_rsrc((void*)(NIN >= 1 ? i0 : (const void*)C), 0, 0x7fffffff, 0x00020000);
    const rsrc_t rs_i1 = __builtin_amdgcn_make_buffer_rsrc((void*)(NIN >= 2 ? i1 : (const void*)C), 0, 0x7fffffff, 0x00020000);
    const rsrc_t rs_i2 = __builtin_amdgcn_make_buffer_rsrc((void*)(NIN >= 3 ? i2 : (const void*)C), 0, 0x7fffffff, 0x00020000);
    const rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(GIN ? (const void*)P.gin : ATT ? (const void*)P.u : (const void*)C), 0, 0x7fffffff, 0x00020000);
    const rsrc_t rs_o0 = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, 0x7fffffff, 0x00020000);
    const rsrc_t rs_o1 = __builtin_amdgcn_make_buffer_rsrc((void*)(NOUT >= 2 ? (void*)out1 : (void*)C), 0, 0x7fffffff, 0x00020000);
    const rsrc_t rs_o2 = __builtin_amdgcn_make_buffer_rsrc((void*)(NOUT >= 3 ? (void*)P.out2 : (void*)C), 0, 0x7fffffff, 0x00020000);
    // PREFETCH DISTANCE.  The tool build's tick sums per epilogue kind (thread 0, configs[4] bf16 step): a pass of a kind WITHOUT input
    // streams (sigmoid z, the attention's t) spends ~1.0 us behind its staging barriers, a pass WITH them 2.1 - 4.5 us -- with the inputs of
    // pass p + 1 requested only one pass (~2.5 us) ahead, every pass still waited for its loads.  Kinds whose inputs are few registers
    // therefore request them further ahead: PD passes, PD + 1 register sets of 2 items x (input streams) x 16 B.
    constexpr int PER_SET = 2 * (NIN + ((GIN || ATT) ? 2 : 0));      // 16-byte registers per set
#ifdef GH_PP_PD
    constexpr int PD = PER_SET == 0 ? 1 : (GH_PP_PD);
#else
    constexpr int PD = PER_SET == 0 ? 1 : PER_SET <= 2 ? 2 : 1;
#endif
    constexpr int NSET = PD + 1;
    pp_u32x4 ra[NSET][2], rb[NSET][2], rc[NSET][2], rg[NSET][2][2];      // [register set][item]; gin: two 16-byte halves of eight fp32
    float4 sw0 = make_float4(0.f, 0.f, 0.f, 0.f), sw1 = sw0;
    const rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc((void*)(SC ? (void*)P.e : (void*)C), 0, 0x7fffffff, 0x00020000);
    if constexpr (SC) {
      const int c0 = 8 * (tid & 31);
      if (c0 < N) { sw0 = *reinterpret_cast<const float4*>(P.w2 + c0); sw1 = *reinterpret_cast<const float4*>(P.w2 + c0 + 4); }
    }
    auto voff = [&](int p_, int j, int esz) __attribute__((always_inline)) {
      int row, col, rr;
      const bool ok = pp_item(p_, j, row, col, rr);
      return ok ? (unsigned)(row * ldc + col) * (unsigned)esz : OOB;
    };
    // EPI_ATT: byte offset of the item's u row (the row's pair / claim through rowg, else row / R), all passes up front
    unsigned uoff[MI][2];
    if constexpr (ATT) {
#pragma unroll
      for (int p_ = 0; p_ < MI; ++p_)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          int row, col, rr;
          const bool ok = pp_item(p_, j, row, col, rr);
          const int rc_ = min(row, M - 1);
          const int ur = P.rowg ? P.rowg[rc_] : rc_ / P.R;
          uoff[p_][j] = ok ? (unsigned)(ur * P.ldu + col) * 4u : OOB;
        }
    }
    auto issue = [&](auto PT, auto SET) __attribute__((always_inline)) {
      constexpr int p_ = decltype(PT)::value, set = decltype(SET)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned vo = voff(p_, j, 2);
        if constexpr (NIN >= 1) ra[set][j] = __builtin_bit_cast(pp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_i0, vo, 0, 0));
        if constexpr (NIN >= 2) rb[set][j] = __builtin_bit_cast(pp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_i1, vo, 0, 0));
        if constexpr (NIN >= 3) rc[set][j] = __builtin_bit_cast(pp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_i2, vo, 0, 0));
        if constexpr (GIN || ATT) {
          unsigned vg;
          if constexpr (ATT) vg = uoff[p_][j]; else vg = voff(p_, j, 4);
          rg[set][j][0] = __builtin_bit_cast(pp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, vg, 0, 0));
          rg[set][j][1] = __builtin_bit_cast(pp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, vg, 16, 0));
        }
      }
    };
    auto st = [&](const rsrc_t rs, unsigned vo, const float4 a, const float4 b) __attribute__((always_inline)) {
      // nontemporal (aux = nt): an epilogue stream is ~100 MB that the next launch reads from its start; its tail in L2 only evicts
      // the weight panels the K loops re-read (A/B on configs[4] bf16, one box: 146.5 -> 148.0 K pairs/s, gemm_big 3.12 -> 3.10 ms;
      // nontemporal only for the backward-only streams r / h~, default policy for z, r xp, out and the gate heads: 3.06 -> 3.11 ms)
      __builtin_amdgcn_raw_buffer_store_b128(pp_pack(a, b), rs, vo, 0, 2);
    };
    auto compute = [&](auto PT) __attribute__((always_inline)) {
      constexpr int p_ = decltype(PT)::value, set = p_ % NSET;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int row, col, rr;
        (void)pp_item(p_, j, row, col, rr);
        const unsigned vo = voff(p_, j, 2);
        float* sp = ep + rr * EP_PITCH + col;
        float4 wa = add4(*reinterpret_cast<const float4*>(sp), *reinterpret_cast<const float4*>(bsum + col));
        float4 wb = add4(*reinterpret_cast<const float4*>(sp + 4), *reinterpret_cast<const float4*>(bsum + col + 4));
        if (E == EPI_ATT) {      // t = tanh(v + u[pair]) -> C (fp32) and back into the staged row for the head-score reduction
          const f32x4 g0 = __builtin_bit_cast(f32x4, rg[set][j][0]), g1 = __builtin_bit_cast(f32x4, rg[set][j][1]);
          const float4 ta = tanh4(add4(wa, make_float4(g0[0], g0[1], g0[2], g0[3])));
          const float4 tb = tanh4(add4(wb, make_float4(g1[0], g1[1], g1[2], g1[3])));
          const unsigned vo4 = voff(p_, j, 4);
          __builtin_amdgcn_raw_buffer_store_b128(pp_u32x4{__builtin_bit_cast(unsigned, ta.x), __builtin_bit_cast(unsigned, ta.y), __builtin_bit_cast(unsigned, ta.z), __builtin_bit_cast(unsigned, ta.w)}, rs_o0, vo4, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(pp_u32x4{__builtin_bit_cast(unsigned, tb.x), __builtin_bit_cast(unsigned, tb.y), __builtin_bit_cast(unsigned, tb.z), __builtin_bit_cast(unsigned, tb.w)}, rs_o0, vo4, 16, 0);
          if (vo != OOB) {
            *reinterpret_cast<float4*>(sp) = ta;
            *reinterpret_cast<float4*>(sp + 4) = tb;
          }
        } else if (E == EPI_STORE) {
          if (drop_mode == 3) {
            const unsigned idx = (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col);
            wa = drop4(wa, drop_seed, idx, drop_thresh, drop_scale);
            wb = drop4(wb, drop_seed, idx + 4u, drop_thresh, drop_scale);
          }
          if constexpr (ACC) { const PF8 x = pp_cvt(ra[set][j]); wa = add4(wa, x.a); wb = add4(wb, x.b); }
          st(rs_o0, vo, wa, wb);
        } else if (E == EPI_SIGMOID_Z) {
          st(rs_o0, vo, sig4(wa), sig4(wb));
        } else if (E == EPI_SIGMOID_R) {
          const PF8 x = pp_cvt(ra[set][j]);
          const float4 r4a = sig4(wa), r4b = sig4(wb);
          st(rs_o0, vo, r4a, r4b);
          st(rs_o1, vo, mul4(r4a, x.a), mul4(r4b, x.b));
        } else if (E == EPI_TANH_H) {
          const PF8 z = pp_cvt(ra[set][j]), x = pp_cvt(rb[set][j]);
          const float4 ha = tanh4(wa), hb = tanh4(wb);
          const float4 ya = mix4(ha, z.a, x.a), yb = mix4(hb, z.b, x.b);
          st(rs_o0, vo, ha, hb);
          st(rs_o1, vo, ya, yb);
          if constexpr (SC) {      // the word scorer sees dropout(out) in fp32 (wrapper.py:189-190)
            float4 sa = ya, sb = yb;
            if (drop_mode == 2) {
              const unsigned idx = (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col);
              sa = drop4(sa, drop_seed, idx, drop_thresh, drop_scale);
              sb = drop4(sb, drop_seed, idx + 4u, drop_thresh, drop_scale);
            }
            float v = vo != OOB ? sa.x * sw0.x + sa.y * sw0.y + sa.z * sw0.z + sa.w * sw0.w + sb.x * sw1.x + sb.y * sw1.y + sb.z * sw1.z + sb.w * sw1.w : 0.f;
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_e, ((lane & 31) == 0 && row < M) ? (unsigned)row * 4u : OOB, 0, 0);
          } else if (scorer) {      // (partials added atomically / whole rows: the staged form, reduced by row_reduce below)
            float4 sa = ya, sb = yb;
            if (drop_mode == 2) {
              const unsigned idx = (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col);
              sa = drop4(sa, drop_seed, idx, drop_thresh, drop_scale);
              sb = drop4(sb, drop_seed, idx + 4u, drop_thresh, drop_scale);
            }
            if (vo != OOB) {
              *reinterpret_cast<float4*>(sp) = sa;
              *reinterpret_cast<float4*>(sp + 4) = sb;
            }
          }
        } else if (E == EPI_BWD_DRX) {
          const PF8 x = pp_cvt(ra[set][j]), r = pp_cvt(rb[set][j]), d = pp_cvt(rc[set][j]);
          st(rs_o0, vo, drx4(wa, x.a, r.a), drx4(wb, x.b, r.b));
          st(rs_o1, vo, add4(d.a, mul4(wa, r.a)), add4(d.b, mul4(wb, r.b)));
        } else if (E == EPI_GATE_PRE) {
          if (drop_mode == 3) {
            const unsigned idx = (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col);
            wa = drop4(wa, drop_seed, idx, drop_thresh, drop_scale);
            wb = drop4(wb, drop_seed, idx + 4u, drop_thresh, drop_scale);
          }
          if constexpr (GIN) {
            const f32x4 g0 = __builtin_bit_cast(f32x4, rg[set][j][0]), g1 = __builtin_bit_cast(f32x4, rg[set][j][1]);
            wa = add4(wa, make_float4(g0[0], g0[1], g0[2], g0[3]));
            wb = add4(wb, make_float4(g1[0], g1[1], g1[2], g1[3]));
          }
          const PF8 Z = pp_cvt(ra[set][j]), Hh = pp_cvt(rb[set][j]), X = pp_cvt(rc[set][j]);
          float4 a0, b0, c0, a1, b1, c1;
          pp_head(wa, Z.a, Hh.a, X.a, a0, b0, c0);
          pp_head(wb, Z.b, Hh.b, X.b, a1, b1, c1);
          st(rs_o0, vo, a0, a1); st(rs_o1, vo, b0, b1); st(rs_o2, vo, c0, c1);
        }
      }
    };
    auto pass = [&](auto PT) __attribute__((always_inline)) {
      constexpr int p_ = decltype(PT)::value;
#ifdef GH_MEASURE
      const bool ticks = (dbg_bits & 1024) != 0;      // GH_DBG=1024: per-kind tick sums (costs atomics: not for timing runs)
      const unsigned long long tp0 = ticks ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
      pp_stage(PT);
#ifdef GH_MEASURE
      const unsigned long long tp1 = ticks ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
      if constexpr (p_ + PD < MI) issue(std::integral_constant<int, p_ + PD>{}, std::integral_constant<int, (p_ + PD) % NSET>{});
      compute(PT);
#ifdef GH_MEASURE
      if (ticks && tid == 0) {
        const unsigned long long tp2 = __builtin_amdgcn_s_memtime();
        const int kslot = (E & 7) + (SC ? 8 : 0);
        atomicAdd(&g_nt_phase[kslot * 4 + 0], (unsigned long long)(tp1 - tp0));
        atomicAdd(&g_nt_phase[kslot * 4 + 1], (unsigned long long)(tp2 - tp1));
        atomicAdd(&g_nt_phase[kslot * 4 + 2], 1ull);
      }
#endif
      if constexpr (E == EPI_TANH_H && !SC) { if (rowred) row_reduce(PT); }
      if constexpr (ATT) row_reduce(PT);
    };
    issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    if constexpr (PD >= 2) issue(std::integral_constant<int, 1>{}, std::integral_constant<int, 1 % NSET>{});
    if constexpr (PD >= 3) issue(std::integral_constant<int, 2>{}, std::integral_constant<int, 2 % NSET>{});
    pass(std::integral_constant<int, 0>{}); pass(std::integral_constant<int, 1>{}); pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{}); pass(std::integral_constant<int, 4>{}); pass(std::integral_constant<int, 5>{});
    pass(std::integral_constant<int, 6>{}); pass(std::integral_constant<int, 7>{});
  };
  {
    constexpr std::integral_constant<bool, false> NO{};
    constexpr std::integral_constant<bool, true> YES{};
    const bool no_c32 = c32 == nullptr;
    if (!(dbg_bits & 128) && no_c32) {
      if (epi == EPI_ATT && pp_io == 0) { pp_fast(std::integral_constant<int, EPI_ATT>{}, NO, NO, NO); return; }
      if (epi == EPI_STORE && (pp_io & 1)) {
        if (accumulate) pp_fast(std::integral_constant<int, EPI_STORE>{}, YES, NO, NO); else pp_fast(std::integral_constant<int, EPI_STORE>{}, NO, NO, NO);
        return;
      }
      if (epi == EPI_SIGMOID_Z && (pp_io & 1)) { pp_fast(std::integral_constant<int, EPI_SIGMOID_Z>{}, NO, NO, NO); return; }
      if (epi == EPI_SIGMOID_R && (pp_io & 7) == 7) { pp_fast(std::integral_constant<int, EPI_SIGMOID_R>{}, NO, NO, NO); return; }
      if (epi == EPI_TANH_H && (pp_io & 15) == 15) {
        if (scorer && P.e_atomic != 1 && !(dbg_bits & 512)) pp_fast(std::integral_constant<int, EPI_TANH_H>{}, NO, NO, YES);
        else pp_fast(std::integral_constant<int, EPI_TANH_H>{}, NO, NO, NO);
        return;
      }
      if (epi == EPI_BWD_DRX && (pp_io & 15) == 15) { pp_fast(std::integral_constant<int, EPI_BWD_DRX>{}, NO, NO, NO); return; }
      if (epi == EPI_GATE_PRE && (pp_io & 63) == 63) {
        if (P.gin) pp_fast(std::integral_constant<int, EPI_GATE_PRE>{}, NO, YES, NO); else pp_fast(std::integral_constant<int, EPI_GATE_PRE>{}, NO, NO, NO);
        return;
      }
    }
  }

  // ------------------------------------------------------------------ generic form: run-time stream formats
  if (epi == EPI_ATT) pp_done = false;
  else {
  const float* s0p = (epi == EPI_SIGMOID_R || epi == EPI_TANH_H || epi == EPI_BWD_DRX || epi == EPI_GATE_PRE) ? in0 : nullptr;
  const float* s1p = (epi == EPI_TANH_H || epi == EPI_BWD_DRX || epi == EPI_GATE_PRE) ? in1 : nullptr;
  const float* s2p = epi == EPI_BWD_DRX ? (const float*)out1 : epi == EPI_GATE_PRE ? P.in2 : (epi == EPI_STORE && accumulate) ? (const float*)C : nullptr;
  const bool s2bf = epi == EPI_BWD_DRX ? (pp_io & 2) != 0 : epi == EPI_GATE_PRE ? (pp_io & 16) != 0 : (pp_io & 1) != 0;
  const bool pf0 = s0p && (pp_io & 4), pf1 = s1p && (pp_io & 8), pf2 = s2p && s2bf;      // streams prefetched as raw bf16
  pp_u32x4 r0[2][2], r1[2][2], r2[2][2];      // [register set][item]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) r0[a][b] = r1[a][b] = r2[a][b] = pp_u32x4{0u, 0u, 0u, 0u};
  auto pp_issue = [&](auto PT, auto SET) __attribute__((always_inline)) {
    constexpr int p_ = decltype(PT)::value, set = decltype(SET)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int row, col, rr;
      if (pp_item(p_, j, row, col, rr)) {
        const size_t o = (size_t)row * ldc + col;
        if (pf0) r0[set][j] = *reinterpret_cast<const pp_u32x4*>(reinterpret_cast<const pp_u16*>(s0p) + o);
        if (pf1) r1[set][j] = *reinterpret_cast<const pp_u32x4*>(reinterpret_cast<const pp_u16*>(s1p) + o);
        if (pf2) r2[set][j] = *reinterpret_cast<const pp_u32x4*>(reinterpret_cast<const pp_u16*>(s2p) + o);
      }
    }
  };
  auto pp_get = [&](const float* p, bool pf, const pp_u32x4 raw, size_t o) __attribute__((always_inline)) {
    if (pf) return pp_cvt(raw);
    PF8 r;
    r.a = *reinterpret_cast<const float4*>(p + o);
    r.b = *reinterpret_cast<const float4*>(p + o + 4);
    return r;
  };
  auto pp_st8 = [&](float* p, size_t o, const float4 a, const float4 b, bool bf) __attribute__((always_inline)) {
    if (bf) {
      *reinterpret_cast<pp_u32x4*>(reinterpret_cast<pp_u16*>(p) + o) = pp_pack(a, b);
    } else {
      *reinterpret_cast<float4*>(p + o) = a;
      *reinterpret_cast<float4*>(p + o + 4) = b;
    }
  };
  auto pp_compute = [&](auto EPI, auto PT) __attribute__((always_inline)) {
    constexpr int E = decltype(EPI)::value, p_ = decltype(PT)::value, set = p_ & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int row, col, rr;
      if (!pp_item(p_, j, row, col, rr)) continue;
      float* sp = ep + rr * EP_PITCH + col;
      float4 wa = add4(*reinterpret_cast<const float4*>(sp), *reinterpret_cast<const float4*>(bsum + col));
      float4 wb = add4(*reinterpret_cast<const float4*>(sp + 4), *reinterpret_cast<const float4*>(bsum + col + 4));
      const size_t o = (size_t)row * ldc + col;
      if (E == EPI_STORE) {
        if (drop_mode == 3) {
          const unsigned idx = (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col);
          wa = drop4(wa, drop_seed, idx, drop_thresh, drop_scale);
          wb = drop4(wb, drop_seed, idx + 4u, drop_thresh, drop_scale);
        }
        if (accumulate) { const PF8 x = pp_get(s2p, pf2, r2[set][j], o); wa = add4(wa, x.a); wb = add4(wb, x.b); }
        pp_st8(C, o, wa, wb, io & 1);
      } else if (E == EPI_SIGMOID_Z) {
        pp_st8(C, o, sig4(wa), sig4(wb), io & 1);
      } else if (E == EPI_SIGMOID_R) {
        const PF8 x = pp_get(s0p, pf0, r0[set][j], o);
        const float4 ra = sig4(wa), rb = sig4(wb);
        pp_st8(C, o, ra, rb, io & 1);
        pp_st8(out1, o, mul4(ra, x.a), mul4(rb, x.b), io & 2);
      } else if (E == EPI_TANH_H) {
        const PF8 z = pp_get(s0p, pf0, r0[set][j], o), x = pp_get(s1p, pf1, r1[set][j], o);
        const float4 ha = tanh4(wa), hb = tanh4(wb);
        const float4 ya = mix4(ha, z.a, x.a), yb = mix4(hb, z.b, x.b);
        pp_st8(C, o, ha, hb, io & 1);
        pp_st8(out1, o, ya, yb, io & 2);
        if (c32) { *reinterpret_cast<float4*>(c32 + o) = ya; *reinterpret_cast<float4*>(c32 + o + 4) = yb; }
        if (scorer) {
          float4 sa = ya, sb = yb;
          if (drop_mode == 2) {
            const unsigned idx = (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col);
            sa = drop4(sa, drop_seed, idx, drop_thresh, drop_scale);
            sb = drop4(sb, drop_seed, idx + 4u, drop_thresh, drop_scale);
          }
          *reinterpret_cast<float4*>(sp) = sa;
          *reinterpret_cast<float4*>(sp + 4) = sb;
        }
      } else if (E == EPI_BWD_DRX) {
        const PF8 x = pp_get(s0p, pf0, r0[set][j], o), r = pp_get(s1p, pf1, r1[set][j], o), d = pp_get(s2p, pf2, r2[set][j], o);
        pp_st8(C, o, drx4(wa, x.a, r.a), drx4(wb, x.b, r.b), io & 1);
        pp_st8(out1, o, add4(d.a, mul4(wa, r.a)), add4(d.b, mul4(wb, r.b)), io & 2);
      } else if (E == EPI_GATE_PRE) {
        if (drop_mode == 3) {
          const unsigned idx = (unsigned)row * (unsigned)drop_ld + (unsigned)(P.drop_col0 + col);
          wa = drop4(wa, drop_seed, idx, drop_thresh, drop_scale);
          wb = drop4(wb, drop_seed, idx + 4u, drop_thresh, drop_scale);
        }
        if (P.gin) { wa = add4(wa, *reinterpret_cast<const float4*>(P.gin + o)); wb = add4(wb, *reinterpret_cast<const float4*>(P.gin + o + 4)); }
        const PF8 Z = pp_get(s0p, pf0, r0[set][j], o), Hh = pp_get(s1p, pf1, r1[set][j], o), X = pp_get(s2p, pf2, r2[set][j], o);
        float4 a0, b0, c0, a1, b1, c1;
        pp_head(wa, Z.a, Hh.a, X.a, a0, b0, c0);
        pp_head(wb, Z.b, Hh.b, X.b, a1, b1, c1);
        pp_st8(C, o, a0, a1, io & 1); pp_st8(out1, o, b0, b1, io & 2); pp_st8(P.out2, o, c0, c1, io & 32);
      }
    }
  };
  auto pp_pass = [&](auto EPI, auto PT) __attribute__((always_inline)) {
    constexpr int p_ = decltype(PT)::value;
    pp_stage(PT);
    if constexpr (p_ + 1 < MI) pp_issue(std::integral_constant<int, p_ + 1>{}, std::integral_constant<int, (p_ + 1) & 1>{});
    pp_compute(EPI, PT);
    if (rowred) row_reduce(PT);
  };
  auto pp_all = [&](auto EPI) __attribute__((always_inline)) {
    pp_issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    pp_pass(EPI, std::integral_constant<int, 0>{}); pp_pass(EPI, std::integral_constant<int, 1>{});
    pp_pass(EPI, std::integral_constant<int, 2>{}); pp_pass(EPI, std::integral_constant<int, 3>{});
    pp_pass(EPI, std::integral_constant<int, 4>{}); pp_pass(EPI, std::integral_constant<int, 5>{});
    pp_pass(EPI, std::integral_constant<int, 6>{}); pp_pass(EPI, std::integral_constant<int, 7>{});
  };
  if (epi == EPI_STORE) pp_all(std::integral_constant<int, EPI_STORE>{});
  else if (epi == EPI_GATE_PRE) pp_all(std::integral_constant<int, EPI_GATE_PRE>{});
  else if (epi == EPI_SIGMOID_Z) pp_all(std::integral_constant<int, EPI_SIGMOID_Z>{});
  else if (epi == EPI_SIGMOID_R) pp_all(std::integral_constant<int, EPI_SIGMOID_R>{});
  else if (epi == EPI_TANH_H) pp_all(std::integral_constant<int, EPI_TANH_H>{});
  else if (epi == EPI_BWD_DRX) pp_all(std::integral_constant<int, EPI_BWD_DRX>{});
  }
}
