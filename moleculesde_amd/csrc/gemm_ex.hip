// gemm_ex.hip — the general fp32 matrix-core GEMM of the library (gfx950, v_mfma_f32_32x32x2_f32).
//
//     C[M,N] (+)= epilogue( A1[M,K1] . B1 + A2[M,K2] . B2 + bias )          (optionally `groups` independent problems)
//
// One kernel covers every dense product the hot path needs that a library GEMM cannot fuse:
//   * forward of nn.Linear on a concatenated input without materialising the concat (two K segments), e.g.
//     embedding_3D(h) + embedding_X(x) of SDE_model_3D_to_2D_node_adj_dense.py:156 as ONE product;
//   * bias + activation (tanh / SiLU / ELU / shifted softplus / ReLU) in the epilogue, optionally only on a column range
//     (stacked projections with different activations) and optionally ALSO storing the pre-activation;
//   * input gradients through an activation: C = (gY . W) * act'(R) with R the saved pre-activation (or output);
//   * accumulation into C (several consumers of one tensor add their gradients in launch order: deterministic);
//   * a per-row mask (the `flags` of the dense score networks);
//   * block-diagonal / per-channel products (`groups`: blockIdx.z with per-group strides).
// B is either nn.Linear's [N][K] (k contiguous) or [K][N] (n contiguous: input gradients, GCN weights stored [in,out]).
//
// Tiling.  256 threads = 4 waves; block tile (32*TM*2) x 64: waves 2 x 2, each TM x 1 MFMA tiles of 32 x 32.  K tile
// 32.  Both operand tiles are STRAIGHT copies of global memory (16-B loads -> 16-B LDS stores, no transpose):
//   * a k-contiguous operand lands as [rows][36] (row stride 36 floats keeps 16-B alignment and makes the operand
//     read -- ONE ds_read_b128 per lane per 4 MFMAs -- conflict free: the 16 lanes the LDS services together start
//     at 16 distinct multiples of 4 banks).  The MFMA sums over its two k lanes, so lane half h may own the k's
//     {8q+4h .. 8q+4h+3} as long as both operands agree: four consecutive floats per lane = one b128 read;
//   * an n-contiguous operand lands as [32 k][68]: its reads are 32 consecutive floats of one k row (conflict free).
// LDS is double buffered: the global loads of tile t+1 are in flight while tile t is multiplied, and ONE barrier per
// K tile separates "everyone has finished reading stage s" from "stage s is overwritten".
// XCD-aware tile order: consecutive tiles of one row strip (same A rows) run on the same XCD, so the strip is fetched
// into that XCD's L2 once.
#include "msde_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GX_BK 32
#define GX_LDK 36   // row stride of a k-contiguous LDS tile
#define GX_LDN 68   // row stride of an n-contiguous LDS tile (64 columns)

// Activations and their derivatives, selected at COMPILE time inside the epilogue (a run-time switch per element
// turned the epilogue into a chain of scalar branches around every store).  Fast forms: v_exp_f32 / v_rcp_f32 /
// v_log_f32, absolute error ~1e-7.  `r` of the derivative is what the forward saved: the OUTPUT y for tanh / ELU /
// ReLU, the PRE-ACTIVATION z for SiLU / shifted softplus.
template <int ACT> __device__ __forceinline__ float gx_act(float z) {
  if (ACT == MSDE_ACT_TANH) return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * z));
  if (ACT == MSDE_ACT_SILU) return z * __frcp_rn(1.f + __expf(-z));
  if (ACT == MSDE_ACT_ELU) return z > 0.f ? z : __expf(z) - 1.f;
  if (ACT == MSDE_ACT_SSP) return (z > 20.f ? z : __logf(1.f + __expf(z))) - 0.6931471805599453f;
  if (ACT == MSDE_ACT_RELU) return fmaxf(z, 0.f);
  return z;
}
template <int ACT> __device__ __forceinline__ float gx_dact(float r) {
  if (ACT == MSDE_ACT_TANH) return 1.f - r * r;
  if (ACT == MSDE_ACT_SILU) { const float s = __frcp_rn(1.f + __expf(-r)); return s * (1.f + r * (1.f - s)); }
  if (ACT == MSDE_ACT_ELU) return r > 0.f ? 1.f : r + 1.f;
  if (ACT == MSDE_ACT_SSP) return __frcp_rn(1.f + __expf(-r));
  if (ACT == MSDE_ACT_RELU) return r > 0.f ? 1.f : 0.f;
  return 1.f;
}

// Epilogue of one wave: TM accumulator tiles -> C (and Z).  C/D map of the 32x32 MFMA: col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  PLAIN: no derivative / row mask / alpha / accumulation (the forward
// products): nothing but bias, activation and the stores in the unrolled loop.
template <int TM, int ACT, bool PLAIN>
__device__ __forceinline__ void gx_epilogue(const msde_gemm_desc& d, const f32x16 (&acc)[TM], int g, int m_base, int gn,
                                            int lhalf) {
  const float* __restrict__ bias = d.bias ? d.bias + (size_t)g * d.bias_gs : nullptr;
  const float* __restrict__ bias2 = d.bias2 ? d.bias2 + (size_t)g * d.bias_gs : nullptr;
  float* __restrict__ C = d.C + (size_t)g * d.c_gs + gn;
  float* __restrict__ Z = d.Z ? d.Z + (size_t)g * d.c_gs + gn : nullptr;
  const float* __restrict__ R = d.R ? d.R + (size_t)g * d.r_gs + gn : nullptr;
  const float bv = (bias ? bias[gn] : 0.f) + (bias2 ? bias2[gn] : 0.f);
  const bool act_here = ACT != MSDE_ACT_NONE && gn >= d.act_lo && gn < d.act_hi;
  const bool dact = d.epi == MSDE_EPI_DACT, accum = (d.flags & MSDE_GEMM_ACCUMULATE) != 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gm = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
      if (gm < d.M) {
        float v = acc[i][r] + bv;
        if (PLAIN) {
          if (Z) Z[(size_t)gm * d.ldz] = v;
          if (act_here) v = gx_act<ACT>(v);
          C[(size_t)gm * d.ldc] = v;
        } else {
          if (!dact) {
            if (Z) Z[(size_t)gm * d.ldz] = v;
            if (act_here) v = gx_act<ACT>(v);
          } else if (act_here) {
            v *= gx_dact<ACT>(R[(size_t)gm * d.ldr]);
          }
          if (d.rowscale) v *= d.rowscale[gm];
          v *= d.alpha;
          float* dst = C + (size_t)gm * d.ldc;
          if (accum) v += *dst;
          *dst = v;
        }
      }
    }
  }
}

// 4 consecutive floats of a row starting at column k (k % 4 == 0).  Out-of-range elements are NOT zeroed here:
// the load address is clamped into the row and `keep` tells gx_mask4 (applied when the tile goes to LDS, i.e. AFTER
// the MFMAs of the previous tile) what to keep -- zeroing at load time would make the wave wait for the load
// before it starts multiplying.  VEC: the row is 16-B aligned and a float4 is entirely inside or outside [0, kend).
template <bool VEC>
__device__ __forceinline__ float4 gx_ld4(const float* __restrict__ row, int k, int kend, int& keep) {
  float4 v;
  if (VEC) {
    const bool in = k < kend;
    keep = in ? 15 : 0;
    v = *reinterpret_cast<const float4*>(row + (in ? k : 0));
  } else {
    const int rem = kend - k;                                 // elements of this float4 inside the row
    keep = rem >= 4 ? 15 : (rem <= 0 ? 0 : (1 << rem) - 1);
    v.x = row[rem > 0 ? k : 0];
    v.y = row[rem > 1 ? k + 1 : 0];
    v.z = row[rem > 2 ? k + 2 : 0];
    v.w = row[rem > 3 ? k + 3 : 0];
  }
  return v;
}
__device__ __forceinline__ float4 gx_mask4(float4 v, int keep) {
  return make_float4(keep & 1 ? v.x : 0.f, keep & 2 ? v.y : 0.f, keep & 4 ? v.z : 0.f, keep & 8 ? v.w : 0.f);
}

template <int TM, bool B_KM, bool VEC>
__global__ void __launch_bounds__(256)
gemm_ex_kernel(const msde_gemm_desc d) {
  constexpr int BM = 64 * TM, BN = 64;
  constexpr int A_FLOATS = BM * GX_LDK;
  constexpr int B_FLOATS = B_KM ? GX_BK * GX_LDN : BN * GX_LDK;
  __shared__ __attribute__((aligned(16))) float As[2][A_FLOATS];
  __shared__ __attribute__((aligned(16))) float Bs[2][B_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 31, lhalf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware tile order: tiles are numbered row-strip major; block b runs on XCD b % 8, so XCD x takes the
  // contiguous range of tile numbers [x * per, (x+1) * per)
  const int tiles_n = (d.N + BN - 1) / BN, tiles_m = (d.M + BM - 1) / BM;
  const int ntile = tiles_m * tiles_n;
  int b = blockIdx.x;
  {
    const int per = (ntile + 7) >> 3;
    const int t = (b & 7) * per + (b >> 3);
    // tiles beyond ntile (last XCD's range may be short) are handled by blocks whose t >= ntile: they take the
    // leftover slots in natural order
    b = t;
  }
  if (b >= ntile) return;
  const int m0 = (b / tiles_n) * BM, n0 = (b % tiles_n) * BN;
  const int g = blockIdx.y;

  const float* __restrict__ A1 = d.A + (size_t)g * d.a_gs;
  const float* __restrict__ A2 = d.A2 ? d.A2 + (size_t)g * d.a_gs : nullptr;
  const float* __restrict__ B1 = d.B + (size_t)g * d.b_gs;
  const float* __restrict__ B2 = d.B2 ? d.B2 + (size_t)g * d.b_gs : nullptr;
  const int K1 = d.K1, K2 = d.A2 ? d.K2 : 0;
  const int nt1 = (K1 + GX_BK - 1) / GX_BK, nt2 = (K2 + GX_BK - 1) / GX_BK;
  const int ntiles = nt1 + nt2;

  // staging registers: A tile BM x 32 = BM*8 float4 -> 2*TM per thread; B tile 64 x 32 (or 32 x 64) = 512 float4 -> 2.
  // NS register stages: the global loads of tile t+NS are issued while tile t is multiplied, so a load has NS - 1 K
  // tiles of MFMA time to land.  Measured with NS = 4 on the encoders' 3588-row problems: no change (3588x300x600 29.7 us
  // either way) -- those shapes are not latency bound but QUANTISED: 3588 x 300 outputs are 1130 wave tiles of 32 x 32
  // for 1024 SIMDs, so some SIMDs run two and the launch takes two tile times (0.55 efficiency; the library uses
  // 16 x 16 MFMA tiles there).  NS = 2 keeps 48 fewer registers.
  constexpr int NS = 2;
  constexpr int NA = 2 * TM, NB = 2;
  float4 ra[NS][NA], rb[NS][NB];
  int ka[NS][NA], kb_[NS][NB];

  // Staging addresses.  SQ counters showed the staging path issuing ~70 VALU instructions per K tile and wave (64-bit
  // address arithmetic, segment selects, tail clamps and masks) -- 4.4 per MFMA, a quarter of the kernel's time.  So:
  // every thread keeps CONSTANT 32-bit byte offsets relative to a UNIFORM tile base (scalar arithmetic; the loads take
  // the `saddr + voffset` form), and interior tiles (no K tail, no N tail for the k-major layout) take a path without
  // clamps and masks.  Only tail tiles and unaligned operands use the general gx_ld4 / gx_mask4 path.
  unsigned oa1[NA], oa2[NA], ob1[NB], ob2[NB];
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const int idx = p * 256 + tid, r = idx >> 3, kq = (idx & 7) * 4;
    const int gm = min(m0 + r, d.M - 1);                     // rows past M repeat the last row: never stored
    oa1[p] = (unsigned)(((size_t)gm * d.lda + kq) * 4);
    oa2[p] = (unsigned)(((size_t)gm * d.lda2 + kq) * 4);
  }
#pragma unroll
  for (int p = 0; p < NB; ++p) {
    const int idx = p * 256 + tid;
    if (!B_KM) {
      const int gn = min(n0 + (idx >> 3), d.N - 1), kq = (idx & 7) * 4;
      ob1[p] = (unsigned)(((size_t)gn * d.ldb + kq) * 4);
      ob2[p] = (unsigned)(((size_t)gn * d.ldb2 + kq) * 4);
    } else {
      const int kr = idx >> 4, nq = (idx & 15) * 4;
      ob1[p] = (unsigned)(((size_t)kr * d.ldb + n0 + nq) * 4);
      ob2[p] = (unsigned)(((size_t)kr * d.ldb2 + n0 + nq) * 4);
    }
  }
  const bool fast_ok = VEC && d.b_kblk_log2 == 0 && (!B_KM || n0 + BN <= d.N);
  bool fs[NS];                                              // register stage holds an interior tile (no masks)
#pragma unroll
  for (int i = 0; i < NS; ++i) fs[i] = false;

  auto load_tile = [&](int t, float4 (&xa)[NA], float4 (&xb)[NB], int (&ma)[NA], int (&mb)[NB], bool& fast) {
    const bool s2 = t >= nt1;
    const int ldb = s2 ? d.ldb2 : d.ldb, K = s2 ? K2 : K1;
    const int k0 = (s2 ? t - nt1 : t) * GX_BK;
    fast = fast_ok && k0 + GX_BK <= K;
    if (fast) {
      const char* __restrict__ ab = reinterpret_cast<const char*>((s2 ? A2 : A1) + k0);
      const char* __restrict__ bb = reinterpret_cast<const char*>((s2 ? B2 : B1) + (B_KM ? (size_t)k0 * ldb : (size_t)k0));
#pragma unroll
      for (int p = 0; p < NA; ++p) xa[p] = *reinterpret_cast<const float4*>(ab + (s2 ? oa2[p] : oa1[p]));
#pragma unroll
      for (int p = 0; p < NB; ++p) xb[p] = *reinterpret_cast<const float4*>(bb + (s2 ? ob2[p] : ob1[p]));
      return;
    }
    const float* __restrict__ A = s2 ? A2 : A1;
    const float* __restrict__ B = s2 ? B2 : B1;
    const int lda = s2 ? d.lda2 : d.lda;
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int idx = p * 256 + tid, r = idx >> 3, kq = (idx & 7) * 4;
      const int gm = min(m0 + r, d.M - 1);
      xa[p] = gx_ld4<VEC>(A + (size_t)gm * lda, k0 + kq, K, ma[p]);
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int idx = p * 256 + tid;
      if (!B_KM) {                                          // B[n][k]
        const int r = idx >> 3, kq = (idx & 7) * 4;
        const int gn = min(n0 + r, d.N - 1);
        const float* __restrict__ bp = B + (size_t)gn * ldb;
        if (d.b_kblk_log2 > 0) {
          // k is cut into blocks of 2^lg: block q of row n starts at B + q * b_kblk_stride + n * ldb (the stacked
          // [C][F][16] weights of the per-channel GCNs read as one [F][16 C] operand)
          const int kk = k0 + kq, q = kk >> d.b_kblk_log2, rem = kk & ((1 << d.b_kblk_log2) - 1);
          int keep;
          xb[p] = gx_ld4<VEC>(bp + (size_t)q * d.b_kblk_stride, kk < K ? rem : 0, 1 << d.b_kblk_log2, keep);
          mb[p] = kk < K ? keep : 0;
        } else {
          xb[p] = gx_ld4<VEC>(bp, k0 + kq, K, mb[p]);
        }
      } else {                                              // B[k][n]: 16 float4 per k row
        const int kr = idx >> 4, nq = (idx & 15) * 4;
        const int gk = min(k0 + kr, K - 1);
        xb[p] = gx_ld4<VEC>(B + (size_t)gk * ldb, n0 + nq, d.N, mb[p]);
        if (k0 + kr >= K) mb[p] = 0;
      }
    }
  };
  auto store_tile = [&](int s, const float4 (&xa)[NA], const float4 (&xb)[NB], const int (&ma)[NA], const int (&mb)[NB],
                        bool fast) {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int idx = p * 256 + tid, r = idx >> 3, kq = (idx & 7) * 4;
      *reinterpret_cast<float4*>(&As[s][r * GX_LDK + kq]) = fast ? xa[p] : gx_mask4(xa[p], ma[p]);
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int idx = p * 256 + tid;
      if (!B_KM) {
        const int r = idx >> 3, kq = (idx & 7) * 4;
        *reinterpret_cast<float4*>(&Bs[s][r * GX_LDK + kq]) = fast ? xb[p] : gx_mask4(xb[p], mb[p]);
      } else {
        const int kr = idx >> 4, nq = (idx & 15) * 4;
        *reinterpret_cast<float4*>(&Bs[s][kr * GX_LDN + nq]) = fast ? xb[p] : gx_mask4(xb[p], mb[p]);
      }
    }
  };

  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // LDS operand reads are software pipelined by hand: the fragment of step q+1 is requested BEFORE the four MFMAs of
  // step q are issued (the compiler otherwise reuses the fragment registers and exposes one LDS latency per 8 MFMAs)
  auto read_frag = [&](const float* __restrict__ as, const float* __restrict__ bs, int q, float4 (&af)[TM], float4& bf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
      af[i] = *reinterpret_cast<const float4*>(&as[((wm * TM + i) * 32 + lcol) * GX_LDK + 8 * q + 4 * lhalf]);
    if (!B_KM) {
      bf = *reinterpret_cast<const float4*>(&bs[(wn * 32 + lcol) * GX_LDK + 8 * q + 4 * lhalf]);
    } else {
      const int kb = 8 * q + 4 * lhalf, c = wn * 32 + lcol;
      bf = make_float4(bs[kb * GX_LDN + c], bs[(kb + 1) * GX_LDN + c], bs[(kb + 2) * GX_LDN + c], bs[(kb + 3) * GX_LDN + c]);
    }
  };
  // a wave whose 32-column half of the tile lies outside N (narrow layers) or whose rows lie outside M only helps with
  // the staging: its SIMD is free for the MFMAs of other workgroups
  const bool wave_live = (n0 + wn * 32 < d.N) && (m0 + wm * TM * 32 < d.M);
  auto compute_tile = [&](int s) {
    if (!wave_live) return;
    const float* __restrict__ as = As[s];
    const float* __restrict__ bs = Bs[s];
    float4 af[2][TM], bf[2];
    read_frag(as, bs, 0, af[0], bf[0]);
#pragma unroll
    for (int q = 0; q < GX_BK / 8; ++q) {                   // k's 8q .. 8q+7: lane half h owns 8q+4h .. 8q+4h+3
      if (q + 1 < GX_BK / 8) read_frag(as, bs, q + 1, af[(q + 1) & 1], bf[(q + 1) & 1]);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][i].x, bf[q & 1].x, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][i].y, bf[q & 1].y, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][i].z, bf[q & 1].z, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][i].w, bf[q & 1].w, acc[i], 0, 0, 0);
      }
    }
  };

  if (ntiles > 0) {
#pragma unroll
    for (int i = 0; i < NS; ++i)
      if (i < ntiles) load_tile(i, ra[i], rb[i], ka[i], kb_[i], fs[i]);
    store_tile(0, ra[0], rb[0], ka[0], kb_[0], fs[0]);
    __syncthreads();
    // step tt (tile tt is in LDS stage tt & 1, its register stage tt % NS is free again): request tile tt + NS into that
    // register stage, multiply tile tt, move tile tt + 1 (register stage (tt + 1) % NS, requested NS - 1 steps ago) into
    // the other LDS stage, which was last read in step tt - 1.  One barrier per K tile.
    const bool dbg_noload = (d.flags & 256) != 0, dbg_nosync = (d.flags & 512) != 0;   // diagnostics (tools/bench_gemm_ex.py)
    for (int t = 0; t < ntiles; t += NS) {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const int tt = t + k;
        if (tt < ntiles) {
          if (tt + NS < ntiles && !dbg_noload) load_tile(tt + NS, ra[k], rb[k], ka[k], kb_[k], fs[k]);
          compute_tile(k & 1);
          const int kn = (k + 1) % NS;                   // compile-time after unrolling
          if (tt + 1 < ntiles && !dbg_noload) store_tile((k + 1) & 1, ra[kn], rb[kn], ka[kn], kb_[kn], fs[kn]);
          if (!dbg_nosync) __syncthreads();
        }
      }
    }
  }

  // ---- epilogue: the activation (and the plain / general form) is chosen ONCE, outside the element loop
  const int gn = n0 + wn * 32 + lcol;
  if (gn < d.N && !(d.flags & 1024)) {
    const int mb = m0 + wm * TM * 32;
    const bool plain = d.epi == MSDE_EPI_ACT && !d.rowscale && d.alpha == 1.f && !(d.flags & MSDE_GEMM_ACCUMULATE);
#define GX_EPI(ACT_)                                                        \
    case ACT_:                                                                \
      if (plain) gx_epilogue<TM, ACT_, true>(d, acc, g, mb, gn, lhalf);       \
      else gx_epilogue<TM, ACT_, false>(d, acc, g, mb, gn, lhalf);            \
      break;
    switch (d.act) {
      GX_EPI(MSDE_ACT_TANH)
      GX_EPI(MSDE_ACT_SILU)
      GX_EPI(MSDE_ACT_ELU)
      GX_EPI(MSDE_ACT_SSP)
      GX_EPI(MSDE_ACT_RELU)
      default:
        if (plain) gx_epilogue<TM, MSDE_ACT_NONE, true>(d, acc, g, mb, gn, lhalf);
        else gx_epilogue<TM, MSDE_ACT_NONE, false>(d, acc, g, mb, gn, lhalf);
    }
#undef GX_EPI
  }
}

// ------------------------------------------------------------------------------------------------ small problems
// C[M,N] = A[M,K] . B (+ bias) for M <= 512 rows (or N <= 32 columns and M <= 8192) and K <= 512 (the MD17 force fine-tuning step: 21 atoms / 420 edges, ~130
// products per step, finetune_MD17.py:47-78).  Such a product is a LATENCY chain, not a throughput problem: the tiled kernel
// above walks K in 32-wide steps behind barriers with ONE 32 x 32 accumulator per wave (K = 300: 150 dependent 64-cycle
// MFMAs + 10 load -> LDS -> barrier round trips, ~10 us).  Here a workgroup owns a 32 x 32 output tile, its four waves
// split K, every wave requests ALL of its operand fragments at once straight into MFMA operand registers (no LDS staging: the
// k order inside a 16-wide chunk is free as long as A and B agree, so lane group g takes k = 4g .. 4g+3 as one 16-byte
// load), runs its v_mfma_f32_16x16x4_f32 on four independent accumulators, and the four partial tiles meet in LDS.
typedef float gs_f4 __attribute__((ext_vector_type(4)));
#define GS_CH 8                 // 16-wide K chunks per wave: K <= 4 * 8 * 16 = 512

template <bool B_KM, bool VEC>
__global__ void __launch_bounds__(256)
gemm_small_kernel(const msde_gemm_desc d) {
  __shared__ float part[4][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  const int tiles_n = (d.N + 31) >> 5;
  const int m0 = ((int)blockIdx.x / tiles_n) * 32, n0 = ((int)blockIdx.x % tiles_n) * 32;
  const int K = d.K1;
  const int nchunk = (K + 15) >> 4, cpw = (nchunk + 3) >> 2;
  const int c_lo = wave * cpw, c_hi = min(c_lo + cpw, nchunk);
  const int grp = blockIdx.y;                                  // grouped products: group strides as in the tiled kernel
  const float* __restrict__ Ag = d.A + (size_t)grp * d.a_gs;
  const float* __restrict__ Bg = d.B + (size_t)grp * d.b_gs;
  // rows / columns past the edge repeat the last one: their results are never stored
  const float* __restrict__ arow[2];
  const float* __restrict__ brow[2];
  int bcol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    arow[i] = Ag + (size_t)min(m0 + 16 * i + c, d.M - 1) * d.lda;
    bcol[i] = min(n0 + 16 * i + c, d.N - 1);
    brow[i] = Bg + (B_KM ? (size_t)bcol[i] : (size_t)bcol[i] * d.ldb);
  }
  auto ld_k = [&](const float* __restrict__ row, int k) -> float4 {          // 4 consecutive k of a k-contiguous row
    if (VEC) return k < K ? *reinterpret_cast<const float4*>(row + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    return make_float4(k < K ? row[k] : 0.f, k + 1 < K ? row[k + 1] : 0.f, k + 2 < K ? row[k + 2] : 0.f, k + 3 < K ? row[k + 3] : 0.f);
  };
  float4 av[GS_CH][2], bv[GS_CH][2];
#pragma unroll
  for (int t = 0; t < GS_CH; ++t) {
    const int k = 16 * (c_lo + t) + 4 * g;
    if (c_lo + t < c_hi) {                                   // (uniform in the wave)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[t][i] = ld_k(arow[i], k);
        if (!B_KM) {
          bv[t][i] = ld_k(brow[i], k);
        } else {
          const float* __restrict__ bp = brow[i] + (size_t)k * d.ldb;
          bv[t][i] = make_float4(k < K ? bp[0] : 0.f, k + 1 < K ? bp[d.ldb] : 0.f, k + 2 < K ? bp[2 * (size_t)d.ldb] : 0.f,
                                 k + 3 < K ? bp[3 * (size_t)d.ldb] : 0.f);
        }
      }
    }
  }
  gs_f4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = gs_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < GS_CH; ++t) {
    if (c_lo + t < c_hi) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][i].x, bv[t][j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][i].y, bv[t][j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][i].z, bv[t][j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][i].w, bv[t][j].w, acc[i][j], 0, 0, 0);
        }
    }
  }
  // C/D map of the 16 x 16 MFMA: column = lane & 15, row = 4 (lane >> 4) + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave][(16 * i + 4 * g + r) * 33 + 16 * j + c] = acc[i][j][r];
  __syncthreads();
  // plain products (the MD17 chain's ~130 per step): bias and store, nothing else evaluated
  const bool accum = (d.flags & MSDE_GEMM_ACCUMULATE) != 0;
  if (d.groups == 1 && d.act == MSDE_ACT_NONE && d.epi == MSDE_EPI_ACT && !d.rowscale && d.alpha == 1.f && !d.Z && !d.bias2 &&
      !accum) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = q * 256 + tid, row = idx >> 5, col = idx & 31;
      const int gm = m0 + row, gn = n0 + col;
      if (gm < d.M && gn < d.N) {
        const int o = row * 33 + col;
        float v = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
        if (d.bias) v += d.bias[gn];
        d.C[(size_t)gm * d.ldc + gn] = v;
      }
    }
    return;
  }
  // the tiled kernel's whole epilogue (gx_epilogue: biases, pre-activation store, activation on a column range or the
  // derivative of a saved one, row mask, alpha, accumulation); the activation is a run-time switch here -- one uniform branch
  // per element of a kernel that is a latency chain anyway
  const float* __restrict__ bias = d.bias ? d.bias + (size_t)grp * d.bias_gs : nullptr;
  const float* __restrict__ bias2 = d.bias2 ? d.bias2 + (size_t)grp * d.bias_gs : nullptr;
  float* __restrict__ Cg = d.C + (size_t)grp * d.c_gs;
  float* __restrict__ Zg = d.Z ? d.Z + (size_t)grp * d.c_gs : nullptr;
  const float* __restrict__ Rg = d.R ? d.R + (size_t)grp * d.r_gs : nullptr;
  const bool dact = d.epi == MSDE_EPI_DACT;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = q * 256 + tid, row = idx >> 5, col = idx & 31;
    const int gm = m0 + row, gn = n0 + col;
    if (gm < d.M && gn < d.N) {
      const int o = row * 33 + col;
      float v = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
      v += (bias ? bias[gn] : 0.f) + (bias2 ? bias2[gn] : 0.f);
      const bool act_here = d.act != MSDE_ACT_NONE && gn >= d.act_lo && gn < d.act_hi;
      if (!dact) {
        if (Zg) Zg[(size_t)gm * d.ldz + gn] = v;
        if (act_here) {
          switch (d.act) {
            case MSDE_ACT_TANH: v = gx_act<MSDE_ACT_TANH>(v); break;
            case MSDE_ACT_SILU: v = gx_act<MSDE_ACT_SILU>(v); break;
            case MSDE_ACT_ELU: v = gx_act<MSDE_ACT_ELU>(v); break;
            case MSDE_ACT_SSP: v = gx_act<MSDE_ACT_SSP>(v); break;
            case MSDE_ACT_RELU: v = gx_act<MSDE_ACT_RELU>(v); break;
            default: break;
          }
        }
      } else if (act_here) {
        const float r = Rg[(size_t)gm * d.ldr + gn];
        switch (d.act) {
          case MSDE_ACT_TANH: v *= gx_dact<MSDE_ACT_TANH>(r); break;
          case MSDE_ACT_SILU: v *= gx_dact<MSDE_ACT_SILU>(r); break;
          case MSDE_ACT_ELU: v *= gx_dact<MSDE_ACT_ELU>(r); break;
          case MSDE_ACT_SSP: v *= gx_dact<MSDE_ACT_SSP>(r); break;
          case MSDE_ACT_RELU: v *= gx_dact<MSDE_ACT_RELU>(r); break;
          default: break;
        }
      }
      if (d.rowscale) v *= d.rowscale[gm];
      v *= d.alpha;
      float* dst = Cg + (size_t)gm * d.ldc + gn;
      if (accum) v += *dst;
      *dst = v;
    }
  }
}

static inline bool gx_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int msde_gemm_ex(const msde_gemm_desc* desc, void* stream) {
  if (!desc) return MSDE_EINVAL;
  msde_gemm_desc d = *desc;
  if (d.M < 0 || d.N <= 0 || d.K1 <= 0 || !d.A || !d.B || !d.C || d.groups < 1) return MSDE_EINVAL;
  if (d.A2 && (!d.B2 || d.K2 <= 0)) return MSDE_EINVAL;
  if (d.epi == MSDE_EPI_DACT && d.act != MSDE_ACT_NONE && !d.R) return MSDE_EINVAL;
  if (d.M == 0) return 0;
  if (d.act == MSDE_ACT_NONE || d.act_hi <= d.act_lo) { d.act_lo = 0; d.act_hi = d.act == MSDE_ACT_NONE ? 0 : d.N; }
  const bool km = (d.flags & MSDE_GEMM_B_KMAJOR) != 0;
  // vector (16-B) loads per operand: aligned base, group stride and leading dimension, and a contiguous extent % 4
  auto vec_ok = [&](const float* p, long long gs, int ld, int extent) {
    return gx_al16(p) && (gs % 4 == 0) && (ld % 4 == 0) && (extent % 4 == 0);
  };
  if (d.b_kblk_log2 < 0 || d.b_kblk_log2 > 16 || (d.b_kblk_log2 > 0 && (km || d.A2 || d.b_kblk_log2 < 2))) return MSDE_EINVAL;
  bool vec = vec_ok(d.A, d.a_gs, d.lda, d.K1) && vec_ok(d.B, d.b_gs, d.ldb, km ? d.N : d.K1);
  if (d.b_kblk_log2 > 0) vec = vec && (d.b_kblk_stride % 4 == 0);
  if (d.A2) vec = vec && vec_ok(d.A2, d.a_gs, d.lda2, d.K2) && vec_ok(d.B2, d.b_gs, d.ldb2, km ? d.N : d.K2);
  // the fast staging path keeps per-thread BYTE offsets in 32 bits (relative to a tile base that already contains k0):
  // operands whose extent reaches 4 GiB take the general path with 64-bit addressing
  auto fits32 = [](long long rows, long long ld, long long extra) { return (rows * ld + extra) * 4 < (1LL << 32); };
  if (vec) {
    bool ok = fits32(d.M, d.lda, d.K1) && fits32(km ? d.K1 : d.N, d.ldb, km ? d.N : d.K1);
    if (d.A2) ok = ok && fits32(d.M, d.lda2, d.K2) && fits32(km ? d.K2 : d.N, d.ldb2, km ? d.N : d.K2);
    if (!ok) vec = false;
  }
  // ... and tall products with <= 128 output columns (per group) and N * K <= 48 K, every epilogue: the tiled kernel gives them
  // M / 64 workgroups per column tile that each walk the whole K behind barriers (3588 x 16 x 364: 57 workgroups, 28 us
  // against 5.0 us here; 3588 x 64 x 364: 13.1 vs 5.6; 3588 x 128 x 300: 7.6 us).  Wider outputs stay on the tiled kernels
  // (3588 x 512 x 16: 7.0 vs 8.4 us here; N = K = 300: 11.7 vs 17.3 us)
  const bool small_rows = d.M <= 512 && d.groups == 1 && d.act == MSDE_ACT_NONE && d.epi == MSDE_EPI_ACT && !d.rowscale &&
                          d.alpha == 1.f && !d.Z && !d.bias2 && !(d.flags & ~MSDE_GEMM_B_KMAJOR);
  const bool skinny = d.M <= 8192 && d.N <= 128 && (long long)d.N * d.K1 <= 49152;
  if ((small_rows || skinny) && d.K1 <= 16 * 4 * GS_CH && !d.A2 && d.b_kblk_log2 == 0 &&
      !(d.flags & ~(MSDE_GEMM_B_KMAJOR | MSDE_GEMM_ACCUMULATE))) {
    // a latency chain, not a throughput problem: the small-problem kernel (above)
    const bool v4 = gx_al16(d.A) && d.lda % 4 == 0 && d.K1 % 4 == 0 && d.a_gs % 4 == 0 &&
                    (km || (gx_al16(d.B) && d.ldb % 4 == 0 && d.b_gs % 4 == 0));
    dim3 grid(((d.M + 31) / 32) * ((d.N + 31) / 32), d.groups);
    hipStream_t st = as_stream(stream);
    if (km) { if (v4) MSDE_LAUNCH((gemm_small_kernel<true, true>), grid, dim3(256), 0, st, d); else MSDE_LAUNCH((gemm_small_kernel<true, false>), grid, dim3(256), 0, st, d); }
    else { if (v4) MSDE_LAUNCH((gemm_small_kernel<false, true>), grid, dim3(256), 0, st, d); else MSDE_LAUNCH((gemm_small_kernel<false, false>), grid, dim3(256), 0, st, d); }
    MSDE_CHECK_LAUNCH();
    return 0;
  }
  // tile height: 128 rows when that still gives every CU >= 2 tiles, else 64 (skinny problems need the parallelism)
  const long t128 = (long)((d.M + 127) / 128) * ((d.N + 63) / 64) * d.groups;
  const int force_tm = 0;      // (tile-height override of the round-2 sweeps: 0 = the rule below)
  const int tm = force_tm ? force_tm : (t128 >= 2L * msde_num_cus() ? 2 : 1);
  const int tiles = ((d.M + 64 * tm - 1) / (64 * tm)) * ((d.N + 63) / 64);
  const int grid_x = ((tiles + 7) / 8) * 8;       // whole multiples of 8: every XCD gets the same number of slots
  dim3 grid(grid_x, d.groups);
  hipStream_t st = as_stream(stream);
#define GX_GO(TM_, KM_, V_) MSDE_LAUNCH((gemm_ex_kernel<TM_, KM_, V_>), grid, dim3(256), 0, st, d)
  if (tm == 2) {
    if (km) { if (vec) GX_GO(2, true, true); else GX_GO(2, true, false); }
    else { if (vec) GX_GO(2, false, true); else GX_GO(2, false, false); }
  } else {
    if (km) { if (vec) GX_GO(1, true, true); else GX_GO(1, true, false); }
    else { if (vec) GX_GO(1, false, true); else GX_GO(1, false, false); }
  }
#undef GX_GO
  MSDE_CHECK_LAUNCH();
  return 0;
}
