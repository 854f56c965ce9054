// gemm_rs.hip — row-strip fp32 matrix-core GEMMs (gfx950, v_mfma_f32_16x16x4_f32): the forward and input-gradient
// products of every plain nn.Linear on the hot path, with the normalisation / activation / residual work around them
// fused in front of and behind the product.  Reference call sites: Geom3D/models/molecule_gnn_model.py:17,28-29 (GIN MLP
// + BatchNorm), Geom3D/models/schnet.py:141-148,163-167 (lin1 / lin2 / lin), SDE_model_2D_to_3D.py:264-271 (node_emb,
// edge_2D_emb, input_mlp, coff_mlp, project) -- all torch.addmm / torch.mm (+ F.batch_norm, F.relu) in the reference.
//
// gemm_rsa_kernel: "strip of A in LDS, weights streamed per wave" (see gemm_rs.h for the tiling argument).  The weight
// operand is read as [K][N]: an input-gradient product takes nn.Linear's weight as stored, a forward product its
// transposed copy (hip.wt_cache: one batched transpose per optimiser step).
// Fusions (msde_rs_desc): A transform on load (BatchNorm apply + ReLU; the BatchNorm input-gradient formula), the
// transformed strip optionally written back once (for the weight gradient), bias / activation / activation derivative /
// residual / accumulate in the epilogue, and per-strip column statistics of the result (BatchNorm forward: mean and
// centred second moment; BatchNorm backward: sum g and sum g (z - mean)) finished by msde_bn_fin_fwd / _bwd.
#include "gemm_rs.h"

#include "gemm_rs_epi.h"

// Epilogue of one wave: T tiles x RT row tiles of 16 x 16 accumulators, segment by segment (gemm_rs.h: RsSeg).
template <int RT, int T>
__device__ __forceinline__ void rs_epilogue(const msde_rs_desc& d, f32x4 (&acc)[T][RT], int wcol, int m0, int strip,
                                            int strip_rows) {
  const int n = threadIdx.x & 15;
  constexpr int W0 = RsSeg<T>::W[0], W1 = RsSeg<T>::W[1];
  rs_epi_segment<RT, T, W0, 0>(d, acc, wcol + W0 * n, m0, strip, strip_rows);
  if (RsSeg<T>::NSEG > 1) rs_epi_segment<RT, T, (W1 > 0 ? W1 : 1), (W1 > 0 ? W0 : 0)>(d, acc, wcol + 16 * W0 + W1 * n, m0, strip, strip_rows);
}

// ---- A transforms on load -----------------------------------------------------------------------------------------
// MSDE_RS_AXF_AFFINE  BatchNorm apply (+ ReLU): a = max(z s[k] + t[k], 0)
// MSDE_RS_AXF_BNBWD   BatchNorm input gradient: dz = p[k] g' + w[k] z + u[k] with g' = g gated by the fused ReLU
//                     (z s[k] + t[k] > 0); p = gamma rstd, w = -p c2 rstd, u = p (c2 rstd mean - c1) come from
//                     msde_bn_fin_bwd (c1 = mean g', c2 = mean g' xhat)
// The transformed strip is optionally written back once (A_out, by the workgroup of column split 0): the weight gradient's
// operand.  All 256 threads, 16-B pieces, coalesced along k; zero beyond M / K.
// Every global load of a batch of rows is issued before the first result is used: written as a plain loop (run-time bounds,
// one load -> transform -> store per trip) the strip cost 5 (K = 300) to 12 (K = 600) DEPENDENT memory round trips per
// thread -- several microseconds in front of every product.  Wave w stages rows w, w + 4, ...; lanes cover the 16-B pieces
// q = lane, lane + 64, lane + 128 of a row (K <= 768), two rows per batch.
template <int MODE>
__device__ __forceinline__ void rs_stage_mode(const msde_rs_desc& d, bool writer, float* __restrict__ As, int ld, int m0,
                                              int rows) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int K = d.K, M = d.M, kq = rs_kpad(K) / 4;
  const bool relu = (d.flags & MSDE_RS_AXF_RELU) != 0;
  float* __restrict__ out = writer ? d.A_out : nullptr;
  constexpr int NQ = 3, NR = 2;
  if (kq > 64 * NQ) {          // very long rows: simple loop
    for (int r = wave; r < rows; r += 4) {
      const int gm = m0 + r;
      for (int q = lane; q < kq; q += 64) {
        const int k = 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gm < M && k < K) v = *reinterpret_cast<const float4*>(d.A + (size_t)gm * d.lda + k);
        *reinterpret_cast<float4*>(As + r * ld + k) = v;      // (transforms are not offered for K > 768)
      }
    }
    return;
  }
  // per-column vectors of the transform: once per lane
  float4 x0[NQ], x1[NQ], x2[NQ], x3[NQ], x4[NQ];
  const bool gate = MODE == MSDE_RS_AXF_BNBWD && d.xf3 != nullptr;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int k = 4 * (lane + 64 * j);
    const bool in = k < K;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    x0[j] = (MODE != MSDE_RS_AXF_NONE && in) ? *reinterpret_cast<const float4*>(d.xf0 + k) : zero;
    x1[j] = (MODE != MSDE_RS_AXF_NONE && in) ? *reinterpret_cast<const float4*>(d.xf1 + k) : zero;
    x2[j] = (MODE == MSDE_RS_AXF_BNBWD && in) ? *reinterpret_cast<const float4*>(d.xf2 + k) : zero;
    x3[j] = (gate && in) ? *reinterpret_cast<const float4*>(d.xf3 + k) : zero;
    x4[j] = (gate && in) ? *reinterpret_cast<const float4*>(d.xf4 + k) : zero;
  }
  for (int rb = wave; rb < rows; rb += 4 * NR) {
    float4 v[NR][NQ], z[NR][NQ];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int gm = m0 + rb + 4 * i;
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int k = 4 * (lane + 64 * j);
        const bool ok = rb + 4 * i < rows && gm < M && k < K;
        v[i][j] = ok ? *reinterpret_cast<const float4*>(d.A + (size_t)gm * d.lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == MSDE_RS_AXF_BNBWD)
          z[i][j] = ok ? *reinterpret_cast<const float4*>(d.A2 + (size_t)gm * d.lda2 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = rb + 4 * i, gm = m0 + r;
      if (r >= rows) continue;
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int q = lane + 64 * j, k = 4 * q;
        if (q >= kq) continue;
        float4 a = v[i][j];
        const bool ok = gm < M && k < K;
        if (ok && MODE == MSDE_RS_AXF_AFFINE) {
          a = make_float4(fmaf(a.x, x0[j].x, x1[j].x), fmaf(a.y, x0[j].y, x1[j].y), fmaf(a.z, x0[j].z, x1[j].z),
                          fmaf(a.w, x0[j].w, x1[j].w));
          if (relu) a = make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
        } else if (ok && MODE == MSDE_RS_AXF_BNBWD) {
          const float4 zz = z[i][j];
          if (gate) {
            a.x = fmaf(zz.x, x3[j].x, x4[j].x) > 0.f ? a.x : 0.f;
            a.y = fmaf(zz.y, x3[j].y, x4[j].y) > 0.f ? a.y : 0.f;
            a.z = fmaf(zz.z, x3[j].z, x4[j].z) > 0.f ? a.z : 0.f;
            a.w = fmaf(zz.w, x3[j].w, x4[j].w) > 0.f ? a.w : 0.f;
          }
          a = make_float4(fmaf(x0[j].x, a.x, fmaf(x1[j].x, zz.x, x2[j].x)), fmaf(x0[j].y, a.y, fmaf(x1[j].y, zz.y, x2[j].y)),
                          fmaf(x0[j].z, a.z, fmaf(x1[j].z, zz.z, x2[j].z)), fmaf(x0[j].w, a.w, fmaf(x1[j].w, zz.w, x2[j].w)));
        }
        if (MODE != MSDE_RS_AXF_NONE && out && ok) *reinterpret_cast<float4*>(out + (size_t)gm * d.lda_out + k) = a;
        *reinterpret_cast<float4*>(As + r * ld + k) = a;
      }
    }
  }
}

__device__ __forceinline__ void rs_stage_desc(const msde_rs_desc& d, bool writer, float* __restrict__ As, int ld, int m0,
                                              int rows) {
  if (d.axf == MSDE_RS_AXF_AFFINE) rs_stage_mode<MSDE_RS_AXF_AFFINE>(d, writer, As, ld, m0, rows);
  else if (d.axf == MSDE_RS_AXF_BNBWD) rs_stage_mode<MSDE_RS_AXF_BNBWD>(d, writer, As, ld, m0, rows);
  else rs_stage_mode<MSDE_RS_AXF_NONE>(d, writer, As, ld, m0, rows);
}

// ===================================================================================================================
// RSA: one workgroup = one strip of 16 RT rows x one of `splits` column ranges of 64 T columns; wave w owns the 16 T
// consecutive columns starting at (split * 4 + w) * 16 T.
// ===================================================================================================================
template <int RT, int T>
__global__ void __launch_bounds__(256, 2)
gemm_rsa_kernel(const msde_rs_desc d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int S = d.splits;
  const int strip = blockIdx.x / S, split = blockIdx.x - strip * S;
  const int m0 = strip * RT * 16;
  const int ld = rs_lds_ld(d.K);
  rs_stage_desc(d, split == 0, lds, ld, m0, RT * 16);
  const int wave = threadIdx.x >> 6;
  const int wcol = (split * 4 + wave) * 16 * T;       // may lie beyond N (narrow outputs): such a wave computes zeros
  f32x4 acc[T][RT];
#pragma unroll
  for (int c = 0; c < T; ++c)
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (every strip walks k from block 0: starting strip s at block s mod blocks, so that workgroups launched together read
  // different weight rows, measured no difference on 3588 x {128, 300, 600}^2: the L2 serves the shared rows fine)
  rsa_mma<RT, T, true>(lds, ld, d.B, d.ldb, d.N, d.K, wcol, acc, 0);
  if (wcol >= d.N) return;
  rs_epilogue<RT, T>(d, acc, wcol, m0, strip, RT * 16);
}

static inline bool rs_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ===================================================================================================================
// finishing kernels of the fused BatchNorm: one launch of C / 16 workgroups; 16 lanes per column combine the per-strip
// partials in a fixed order.
// ===================================================================================================================
// Both kernels: LPC lanes per column; lane l owns strips l, l + LPC, ... (the first ones kept in registers: all loads are
// issued before anything is summed), lane partials are combined by a fixed butterfly (bitwise reproducible).
#define FIN_REGS 16           // partials of a lane kept in registers (LPC * FIN_REGS strips); beyond that the tail is walked in a loop
template <int LPC>
__device__ __forceinline__ float fin_lane_sum(float v) {
#pragma unroll
  for (int o = 1; o < LPC; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Forward: strip partials (mean_s, M2_s) with n_s valid rows ->  mean = sum n_s mean_s / n,  M2 = sum M2_s + n_s (mean_s -
// mean)^2 (two passes over the partials, no division per strip).  LPC lanes per column (16: round 3; 64 = a wave per
// column: a quarter of the dependent loads per lane, four times the workgroups -- the kernel is pure latency on the GIN chain).
template <int LPC>
__global__ void __launch_bounds__(256)
bn_fin_fwd_kernel(const float* __restrict__ stats, int strips, int strip_rows, int M, const int* __restrict__ m_valid,
                  int C, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                  float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ scale,
                  float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_rstd) {
  constexpr int PL = LPC == 16 ? FIN_REGS : 4;
  const int mv = m_valid ? min(M, m_valid[0]) : M;
  const int l = threadIdx.x & (LPC - 1), col = blockIdx.x * (256 / LPC) + threadIdx.x / LPC;
  const int cc = min(col, C - 1);
  float pm[PL], pq[PL], pn[PL];
#pragma unroll
  for (int k = 0; k < PL; ++k) {
    const int s = l + LPC * k;
    const bool ok = s < strips;
    const size_t o = (size_t)(ok ? s : 0) * 2 * C + cc;
    pm[k] = stats[o];
    pq[k] = stats[o + C];
    pn[k] = ok ? (float)max(0, min(strip_rows, mv - s * strip_rows)) : 0.f;
  }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < PL; ++k) sum = fmaf(pn[k], pn[k] > 0.f ? pm[k] : 0.f, sum);
  for (int s = l + LPC * PL; s < strips; s += LPC) {
    const float n = (float)max(0, min(strip_rows, mv - s * strip_rows));
    if (n > 0.f) sum = fmaf(n, stats[(size_t)s * 2 * C + cc], sum);
  }
  sum = fin_lane_sum<LPC>(sum);
  const float n = (float)mv;
  const float mean = mv > 0 ? sum / n : 0.f;
  float m2 = 0.f;
#pragma unroll
  for (int k = 0; k < PL; ++k)
    if (pn[k] > 0.f) { const float dl = pm[k] - mean; m2 += pq[k] + pn[k] * dl * dl; }
  for (int s = l + LPC * PL; s < strips; s += LPC) {
    const float ns = (float)max(0, min(strip_rows, mv - s * strip_rows));
    if (ns > 0.f) { const float dl = stats[(size_t)s * 2 * C + cc] - mean; m2 += stats[(size_t)s * 2 * C + C + cc] + ns * dl * dl; }
  }
  m2 = fin_lane_sum<LPC>(m2);
  if (col < C && l == 0) {
    const float var = mv > 0 ? m2 / n : 0.f;
    const float rstd = rsqrtf(var + eps);
    const float gv = gamma ? gamma[col] : 1.f, bv = beta ? beta[col] : 0.f;
    scale[col] = gv * rstd;
    shift[col] = bv - mean * gv * rstd;
    save_mean[col] = mean;
    save_rstd[col] = rstd;
    if (running_mean) {
      const float unbiased = mv > 1 ? m2 / (n - 1.f) : var;
      running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * mean;
      running_var[col] = (1.f - momentum) * running_var[col] + momentum * unbiased;
    }
  }
}

template <int LPC>
__global__ void __launch_bounds__(256)
bn_fin_bwd_kernel(const float* __restrict__ stats, int strips, int M, const int* __restrict__ m_valid, int C,
                  const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                  float* __restrict__ p, float* __restrict__ w, float* __restrict__ u, float* __restrict__ dgamma,
                  float* __restrict__ dbeta) {
  constexpr int PL = LPC == 16 ? FIN_REGS : 4;
  const int mv = m_valid ? min(M, m_valid[0]) : M;
  const int l = threadIdx.x & (LPC - 1), col = blockIdx.x * (256 / LPC) + threadIdx.x / LPC;
  const int cc = min(col, C - 1);
  float pa[PL], pb[PL];
#pragma unroll
  for (int k = 0; k < PL; ++k) {
    const int s = l + LPC * k;
    const bool ok = s < strips;
    const size_t o = (size_t)(ok ? s : 0) * 2 * C + cc;
    pa[k] = ok ? stats[o] : 0.f;
    pb[k] = ok ? stats[o + C] : 0.f;
  }
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int k = 0; k < PL; ++k) { a += pa[k]; b += pb[k]; }
  for (int s = l + LPC * PL; s < strips; s += LPC) { a += stats[(size_t)s * 2 * C + cc]; b += stats[(size_t)s * 2 * C + C + cc]; }
  const float ra = fin_lane_sum<LPC>(a), rb = fin_lane_sum<LPC>(b);
  if (col < C && l == 0) {
    const float rs = rstd[col], mu = mean[col], gv = gamma ? gamma[col] : 1.f;
    const float sum_g = ra, sum_gx = rb * rs;          // sum g', sum g' xhat
    if (dbeta) dbeta[col] = sum_g;
    if (dgamma) dgamma[col] = sum_gx;
    const float inv = mv > 0 ? 1.f / (float)mv : 0.f;
    const float c1 = sum_g * inv, c2 = sum_gx * inv;
    const float pp = gv * rs;
    p[col] = pp;
    w[col] = -pp * c2 * rs;
    u[col] = pp * (c2 * rs * mu - c1);
  }
}

// y[m][c] = max(x[m][c] scale[c] + shift[c], 0 if relu): the BatchNorm apply of a tensor that several consumers read
// (the layer output of the GIN stack); C % 4 == 0.
__global__ void __launch_bounds__(256)
affine_cols_kernel(const float* __restrict__ X, int M, int C, const float* __restrict__ scale,
                   const float* __restrict__ shift, int relu, float* __restrict__ Y) {
  const int cq = C >> 2;
  const long total = (long)M * cq;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % cq) * 4;
    const float4 v = *reinterpret_cast<const float4*>(X + i * 4);
    const float4 s = *reinterpret_cast<const float4*>(scale + c), t = *reinterpret_cast<const float4*>(shift + c);
    float4 y = make_float4(fmaf(v.x, s.x, t.x), fmaf(v.y, s.y, t.y), fmaf(v.z, s.z, t.z), fmaf(v.w, s.w, t.w));
    if (relu) y = make_float4(fmaxf(y.x, 0.f), fmaxf(y.y, 0.f), fmaxf(y.z, 0.f), fmaxf(y.w, 0.f));
    *reinterpret_cast<float4*>(Y + i * 4) = y;
  }
}

// The BatchNorm input gradient as a pass of its own (what MSDE_RS_AXF_BNBWD applies to the A fragments of a product):
//   out[m][c] = p[c] g' + w[c] z[m][c] + u[c],  g' = g[m][c] gated to 0 where z[m][c] xf3[c] + xf4[c] <= 0 (xf3 != NULL),
// rows behind the row bound: 0.  Round 5: the transform inside the product costs 2.8 vector instructions per MFMA on a chip
// whose fp32 MFMA hides none of them: 40 us for the fused 3588 x 600 x 300 product against 5 + 25 us for this
// pass and the plain product.  C % 4 == 0, 16-byte aligned rows.
__global__ void __launch_bounds__(256)
bn_bwd_cols_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ Z, int ldz, const float* __restrict__ p,
                   const float* __restrict__ w, const float* __restrict__ u, const float* __restrict__ xf3,
                   const float* __restrict__ xf4, int M, const int* __restrict__ m_valid, int C, float* __restrict__ out, int ldo) {
  const int cq = C >> 2;
  const long total = (long)M * cq;
  const int mv = m_valid ? min(M, m_valid[0]) : M;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i / cq), c = (int)(i % cq) * 4;
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < mv) {
      float4 g = *reinterpret_cast<const float4*>(G + (size_t)m * ldg + c);
      const float4 z = *reinterpret_cast<const float4*>(Z + (size_t)m * ldz + c);
      if (xf3) {
        const float4 a = *reinterpret_cast<const float4*>(xf3 + c), b = *reinterpret_cast<const float4*>(xf4 + c);
        g.x = fmaf(z.x, a.x, b.x) <= 0.f ? 0.f : g.x; g.y = fmaf(z.y, a.y, b.y) <= 0.f ? 0.f : g.y;
        g.z = fmaf(z.z, a.z, b.z) <= 0.f ? 0.f : g.z; g.w = fmaf(z.w, a.w, b.w) <= 0.f ? 0.f : g.w;
      }
      const float4 pp = *reinterpret_cast<const float4*>(p + c), ww = *reinterpret_cast<const float4*>(w + c),
                   uu = *reinterpret_cast<const float4*>(u + c);
      y = make_float4(fmaf(pp.x, g.x, fmaf(ww.x, z.x, uu.x)), fmaf(pp.y, g.y, fmaf(ww.y, z.y, uu.y)),
                      fmaf(pp.z, g.z, fmaf(ww.z, z.z, uu.z)), fmaf(pp.w, g.w, fmaf(ww.w, z.w, uu.w)));
    }
    *reinterpret_cast<float4*>(out + (size_t)m * ldo + c) = y;
  }
}

// bn_fin_bwd + bn_bwd_cols in ONE launch: a workgroup owns 64 columns x a chunk of rows; it first sums the strip partials of
// ITS columns (strips x 2 x 64 values: thread = (column, strip quarter), the four quarters meet in LDS in a fixed order) into
// p | w | u exactly as bn_fin_bwd does, then streams its rows.  The finish was a 5 us launch of its own on the GIN chain in
// front of every column pass (ten per step); recomputing it per workgroup costs ~30 independent loads per thread.  The row
// chunk 0 of a column block also writes dgamma / dbeta.  Fixed summation order (bit-reproducible), not the order of
// bn_fin_bwd<64> (a wave per column).
__global__ void __launch_bounds__(256)
bn_bwd_fin_cols_kernel(const float* __restrict__ stats, int strips, const float* __restrict__ gamma, const float* __restrict__ mean,
                       const float* __restrict__ rstd, const float* __restrict__ G, int ldg, const float* __restrict__ Z, int ldz,
                       const float* __restrict__ xf3, const float* __restrict__ xf4, int M, const int* __restrict__ m_valid, int C,
                       int rows_per_wg, float* __restrict__ out, int ldo, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float pa[4][64], pb[4][64];
  __shared__ float vp[64], vw[64], vu[64];
  const int mv = m_valid ? min(M, m_valid[0]) : M;
  const int c0 = blockIdx.x * 64;
  {
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int cc = min(c0 + cl, C - 1);
    float a = 0.f, b = 0.f;
    for (int s0 = q; s0 < strips; s0 += 64) {          // sixteen strips of this quarter per trip (all 57 strips of the GIN layers in
      float xa[16], xb[16];                            // ONE trip): 32 independent loads in flight, then summed in strip order
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int sidx = s0 + 4 * k;
        const bool ok = sidx < strips;
        const size_t o = (size_t)(ok ? sidx : 0) * 2 * C + cc;
        xa[k] = ok ? stats[o] : 0.f;
        xb[k] = ok ? stats[o + C] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) { a += xa[k]; b += xb[k]; }
    }
    pa[q][cl] = a;
    pb[q][cl] = b;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cl = threadIdx.x, col = c0 + cl;
    if (col < C) {
      const float ra = ((pa[0][cl] + pa[1][cl]) + pa[2][cl]) + pa[3][cl];
      const float rb = ((pb[0][cl] + pb[1][cl]) + pb[2][cl]) + pb[3][cl];
      const float rs = rstd[col], mu = mean[col], gv = gamma ? gamma[col] : 1.f;
      const float sum_g = ra, sum_gx = rb * rs;          // sum g', sum g' xhat
      if (blockIdx.y == 0) {
        if (dbeta) dbeta[col] = sum_g;
        if (dgamma) dgamma[col] = sum_gx;
      }
      const float inv = mv > 0 ? 1.f / (float)mv : 0.f;
      const float c1 = sum_g * inv, c2 = sum_gx * inv;
      const float pp = gv * rs;
      vp[cl] = pp;
      vw[cl] = -pp * c2 * rs;
      vu[cl] = pp * (c2 * rs * mu - c1);
    } else {
      vp[cl] = 0.f; vw[cl] = 0.f; vu[cl] = 0.f;
    }
  }
  __syncthreads();
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = c0 + 4 * cq;
  if (c >= C) return;
  const float4 pp = *reinterpret_cast<const float4*>(&vp[4 * cq]), ww = *reinterpret_cast<const float4*>(&vw[4 * cq]),
               uu = *reinterpret_cast<const float4*>(&vu[4 * cq]);
  float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = ga;
  if (xf3) { ga = *reinterpret_cast<const float4*>(xf3 + c); gb = *reinterpret_cast<const float4*>(xf4 + c); }
  const int r0 = blockIdx.y * rows_per_wg, r1 = min(r0 + rows_per_wg, M);
  for (int m = r0 + rl; m < r1; m += 16) {
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < mv) {
      float4 g = *reinterpret_cast<const float4*>(G + (size_t)m * ldg + c);
      const float4 z = *reinterpret_cast<const float4*>(Z + (size_t)m * ldz + c);
      if (xf3) {
        g.x = fmaf(z.x, ga.x, gb.x) <= 0.f ? 0.f : g.x; g.y = fmaf(z.y, ga.y, gb.y) <= 0.f ? 0.f : g.y;
        g.z = fmaf(z.z, ga.z, gb.z) <= 0.f ? 0.f : g.z; g.w = fmaf(z.w, ga.w, gb.w) <= 0.f ? 0.f : g.w;
      }
      y = make_float4(fmaf(pp.x, g.x, fmaf(ww.x, z.x, uu.x)), fmaf(pp.y, g.y, fmaf(ww.y, z.y, uu.y)),
                      fmaf(pp.z, g.z, fmaf(ww.z, z.z, uu.z)), fmaf(pp.w, g.w, fmaf(ww.w, z.w, uu.w)));
    }
    *reinterpret_cast<float4*>(out + (size_t)m * ldo + c) = y;
  }
}

// BatchNorm-backward partial sums of a gradient that does NOT come out of one of the products above (the gradient of the
// GIN layer output): per 64-row strip and column, sum g' and sum g' (z - mean[c]) with g' = g gated by the fused ReLU
// (y[m][c] > 0, y = the layer output) -- the [strips][2][C] format of MSDE_RS_STATS_BNBWD.  One workgroup per
// (64 columns, strip); 16 row lanes x 16 float4 column groups; C % 4 == 0.
__global__ void __launch_bounds__(256)
bn_bwd_colstats_kernel(const float* __restrict__ G, const float* __restrict__ Z, const float* __restrict__ Y,
                       const float* __restrict__ mean, int M, const int* __restrict__ m_valid, int C,
                       float* __restrict__ stats) {
  __shared__ float sa[16][64], sb[16][64];
  const int mv = m_valid ? min(M, m_valid[0]) : M;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + tx * 4, strip = blockIdx.y;
  const int r0 = strip * 64, r1 = min(r0 + 64, mv);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (c < C) {
    const float4 mu = *reinterpret_cast<const float4*>(mean + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + ty + 16 * k;
      if (r < r1) {
        float4 g = *reinterpret_cast<const float4*>(G + (size_t)r * C + c);
        const float4 z = *reinterpret_cast<const float4*>(Z + (size_t)r * C + c);
        if (Y) {
          const float4 y = *reinterpret_cast<const float4*>(Y + (size_t)r * C + c);
          g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f; g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
        }
        a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
        b.x = fmaf(g.x, z.x - mu.x, b.x); b.y = fmaf(g.y, z.y - mu.y, b.y);
        b.z = fmaf(g.z, z.z - mu.z, b.z); b.w = fmaf(g.w, z.w - mu.w, b.w);
      }
    }
  }
  sa[ty][tx * 4] = a.x; sa[ty][tx * 4 + 1] = a.y; sa[ty][tx * 4 + 2] = a.z; sa[ty][tx * 4 + 3] = a.w;
  sb[ty][tx * 4] = b.x; sb[ty][tx * 4 + 1] = b.y; sb[ty][tx * 4 + 2] = b.z; sb[ty][tx * 4 + 3] = b.w;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc = blockIdx.x * 64 + threadIdx.x;
    if (cc < C) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int l = 0; l < 16; ++l) { s0 += sa[l][threadIdx.x]; s1 += sb[l][threadIdx.x]; }
      stats[(size_t)strip * 2 * C + cc] = s0;
      stats[(size_t)strip * 2 * C + C + cc] = s1;
    }
  }
}

// ===================================================================================================================
// host side
// ===================================================================================================================

// geometry of the RSA kernel for (M, N, K): row tiles per strip (RT), tiles per wave (T), column splits (S = workgroups per
// strip, each covering 64 T columns)
static void rsa_geometry(int M, int N, int K, int* rt, int* tw, int* splits) {
  const int ntiles = (N + 15) / 16;
  const int cus = msde_num_cus();
  int T = (ntiles + 3) / 4;                            // one workgroup per strip if that is <= 5 tiles per wave
  int S = 1;
  if (T > 5) { S = (ntiles + 19) / 20; T = (ntiles + 4 * S - 1) / (4 * S); }
  int RT = 1;
  // 32-row strips halve the weight traffic per MFMA; taken when they still give every CU a workgroup and two strips fit in
  // a CU's LDS
  if ((long)((M + 31) / 32) * S >= (long)(cus * 7) / 8 && (size_t)32 * rs_lds_ld(K) * 4 <= 80 * 1024) RT = 2;
  const int f_rt = 0;         // (geometry overrides of the round-3 sweeps, profiles/r03_gemm_rs_geometry_sweep.txt: 0 = the rule below)
  const int f_t = 0;
  if (f_rt) RT = f_rt;
  if (f_t >= 1 && f_t <= 5) { T = f_t; S = (ntiles + 4 * T - 1) / (4 * T); }
  *rt = RT;
  *tw = T;
  *splits = S;
}

extern "C" int msde_gemm_rs_geometry(int M, int N, int K, int* strips, int* strip_rows) {
  if (M < 0 || N <= 0 || K <= 0 || !strips || !strip_rows) return MSDE_EINVAL;
  int rt, t, s;
  rsa_geometry(M, N, K, &rt, &t, &s);
  *strip_rows = 16 * rt;
  *strips = (M + 16 * rt - 1) / (16 * rt);
  return 0;
}

// dynamic LDS beyond 64 KB must be granted per kernel; remembered per kernel address (the instantiations share one
// function-pointer type, so a static inside the template would be shared too)
#include <map>
#include <mutex>
template <typename KERN>
static int rs_launch(KERN kern, dim3 grid, size_t lds, hipStream_t st, const msde_rs_desc& d) {
  if (lds > 64 * 1024) {
    static std::map<const void*, size_t> granted;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    size_t& gr = granted[reinterpret_cast<const void*>(kern)];
    if (lds > gr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      gr = lds;
    }
  }
  MSDE_LAUNCH(kern, grid, dim3(256), lds, st, d);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_gemm_rs(const msde_rs_desc* desc, void* stream) {
  if (!desc) return MSDE_EINVAL;
  msde_rs_desc d = *desc;
  if (d.M < 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.B || !d.C) return MSDE_EINVAL;
  if (d.M == 0) return 0;
  if (!(d.flags & MSDE_GEMM_B_KMAJOR)) return MSDE_EUNSUP;          // weights are read as [K][N] only (gemm_rs.h)
  // 16-B pieces: K % 4 and an aligned A for the strip; N % 4 and aligned weight rows for the interleaved column segments
  if (d.K % 4 || d.lda % 4 || !rs_al16(d.A)) return MSDE_EUNSUP;
  if ((size_t)d.K * (size_t)d.ldb >= (1u << 30)) return MSDE_EUNSUP;       // 32-bit element offsets of B
  if (d.epi == MSDE_EPI_DACT && d.act != MSDE_ACT_NONE && !d.R) return MSDE_EINVAL;
  if (d.axf != MSDE_RS_AXF_NONE && d.K > 768) return MSDE_EUNSUP;          // transforms are staged with <= 3 pieces per lane
  if (d.axf == MSDE_RS_AXF_AFFINE && (!d.xf0 || !d.xf1 || !rs_al16(d.xf0) || !rs_al16(d.xf1))) return MSDE_EINVAL;
  if (d.axf == MSDE_RS_AXF_BNBWD && (!d.A2 || !d.xf0 || !d.xf1 || !d.xf2 || d.lda2 % 4 || !rs_al16(d.A2))) return MSDE_EINVAL;
  if (d.A_out && (d.lda_out % 4 || !rs_al16(d.A_out))) return MSDE_EINVAL;
  if (d.stats && d.stats_mode == MSDE_RS_STATS_BNBWD && (!d.stats_z || !d.stats_mean)) return MSDE_EINVAL;
  int rt, T, S;
  rsa_geometry(d.M, d.N, d.K, &rt, &T, &S);
  if (d.rt) rt = d.rt;
  if (T > 1 && (d.N % 4 || d.ldb % 4 || !rs_al16(d.B))) return MSDE_EUNSUP;
  size_t lds = (size_t)16 * rt * rs_lds_ld(d.K) * 4;
  if (lds > 160 * 1024 && rt == 2) { rt = 1; lds /= 2; }
  if (lds > 160 * 1024) return MSDE_EUNSUP;
  d.splits = S;
  d.rt = rt;
  d.flags &= ~MSDE_RS_VEC_STORE;
  auto rows_ok = [](const void* p, int ldx) { return !p || (ldx % 4 == 0 && rs_al16(p)); };
  if (rows_ok(d.C, d.ldc) && rows_ok(d.Res, d.ldres) && rows_ok(d.R, d.ldr) && rows_ok(d.Z, d.ldz) &&
      rows_ok(d.stats_z, d.ld_sz) && rows_ok(d.bias, 0) && rows_ok(d.stats_mean, 0) && rows_ok(d.stats, 0) && d.N % 4 == 0)
    d.flags |= MSDE_RS_VEC_STORE;
  dim3 grid(((d.M + 16 * rt - 1) / (16 * rt)) * S);
  hipStream_t st = as_stream(stream);
#define RSA_GO(T_) (rt == 2 ? rs_launch(gemm_rsa_kernel<2, T_>, grid, lds, st, d) : rs_launch(gemm_rsa_kernel<1, T_>, grid, lds, st, d))
  switch (T) {
    case 1: return RSA_GO(1);
    case 2: return RSA_GO(2);
    case 3: return RSA_GO(3);
    case 4: return RSA_GO(4);
    case 5: return RSA_GO(5);
    default: return MSDE_EUNSUP;
  }
#undef RSA_GO
}

// ---- re-laid-out weight copies (forward products read [K][N], gemm_rs.h; stacked / permuted operands of fused layers):
// ONE launch for any number of blocks.  table rows (long long x 8): {src, dst, rows, cols, src_ld, dst_ld, mode, 0} --
// mode 0: dst[c * dst_ld + r] = src[r * src_ld + c] (transpose of a rows x cols block), mode 1: dst[r * dst_ld + c] =
// src[r * src_ld + c] (copy of the block); prefix[i] = first tile of block i, prefix[n] = total.  A tile is 32 x 32 source
// elements.
__device__ __forceinline__ void relayout_tile(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols,
                                              int src_ld, int dst_ld, int mode, int t, float (*tile)[33]) {
  const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
  const int tc = (cols + 31) >> 5;
  const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
  if (mode == 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + y + 8 * k, c = c0 + x;
      if (r < rows && c < cols) dst[(size_t)r * dst_ld + c] = src[(size_t)r * src_ld + c];
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + y + 8 * k, c = c0 + x;
    tile[y + 8 * k][x] = (r < rows && c < cols) ? src[(size_t)r * src_ld + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + y + 8 * k, r = r0 + x;
    if (c < cols && r < rows) dst[(size_t)c * dst_ld + r] = tile[x][y + 8 * k];
  }
}

__global__ void __launch_bounds__(256)
transpose_multi_kernel(const long long* __restrict__ table, const int* __restrict__ prefix, int n) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n - 1;                              // last block whose first tile is <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (prefix[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* e = table + 8 * (size_t)lo;
  relayout_tile(reinterpret_cast<const float*>(e[0]), reinterpret_cast<float*>(e[1]), (int)e[2], (int)e[3], (int)e[4],
                (int)e[5], (int)e[6], blockIdx.x - prefix[lo], tile);
}

__global__ void __launch_bounds__(256)
transpose_one_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols, int src_ld, int dst_ld,
                     int mode) {
  __shared__ float tile[32][33];
  relayout_tile(src, dst, rows, cols, src_ld, dst_ld, mode, blockIdx.x, tile);
}

extern "C" int msde_relayout(const float* src, int src_ld, float* dst, int dst_ld, int rows, int cols, int mode, void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  if (!src || !dst || src_ld < cols || (mode != 0 && mode != 1) || dst_ld < (mode == 1 ? cols : rows)) return MSDE_EINVAL;
  MSDE_LAUNCH(transpose_one_kernel, dim3(((rows + 31) / 32) * ((cols + 31) / 32)), dim3(256), 0, as_stream(stream), src, dst,
              rows, cols, src_ld, dst_ld, mode);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_transpose(const float* src, float* dst, int rows, int cols, void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  if (!src || !dst) return MSDE_EINVAL;
  MSDE_LAUNCH(transpose_one_kernel, dim3(((rows + 31) / 32) * ((cols + 31) / 32)), dim3(256), 0, as_stream(stream), src, dst,
              rows, cols, cols, rows, 0);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_transpose_multi(const long long* table, const int* prefix, int n, int total_tiles, void* stream) {
  if (n <= 0 || total_tiles <= 0) return 0;
  if (!table || !prefix) return MSDE_EINVAL;
  MSDE_LAUNCH(transpose_multi_kernel, dim3(total_tiles), dim3(256), 0, as_stream(stream), table, prefix, n);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_bn_fin_fwd(const float* stats, int strips, int strip_rows, int M, const int* m_valid, int C,
                               const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* scale, float* shift, float* save_mean, float* save_rstd,
                               void* stream) {
  if (!stats || strips <= 0 || strip_rows <= 0 || C <= 0 || !scale || !shift || !save_mean || !save_rstd) return MSDE_EINVAL;
  const int lpc = 64;     // a wave per column (round 4: 2.673 -> 2.652 ms against 16 lanes x 14 partials)
  if (lpc == 16)
    MSDE_LAUNCH(bn_fin_fwd_kernel<16>, dim3((C + 15) / 16), dim3(256), 0, as_stream(stream), stats, strips, strip_rows, M,
                m_valid, C, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_rstd);
  else
    MSDE_LAUNCH(bn_fin_fwd_kernel<64>, dim3((C + 3) / 4), dim3(256), 0, as_stream(stream), stats, strips, strip_rows, M,
                m_valid, C, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_rstd);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_bn_fin_bwd(const float* stats, int strips, int M, const int* m_valid, int C, const float* gamma,
                               const float* mean, const float* rstd, float* p, float* w, float* u, float* dgamma,
                               float* dbeta, void* stream) {
  if (!stats || strips <= 0 || C <= 0 || !mean || !rstd || !p || !w || !u) return MSDE_EINVAL;
  const int lpc = 64;     // a wave per column (round 4: 2.673 -> 2.652 ms against 16 lanes x 14 partials)
  if (lpc == 16)
    MSDE_LAUNCH(bn_fin_bwd_kernel<16>, dim3((C + 15) / 16), dim3(256), 0, as_stream(stream), stats, strips, M, m_valid, C,
                gamma, mean, rstd, p, w, u, dgamma, dbeta);
  else
    MSDE_LAUNCH(bn_fin_bwd_kernel<64>, dim3((C + 3) / 4), dim3(256), 0, as_stream(stream), stats, strips, M, m_valid, C,
                gamma, mean, rstd, p, w, u, dgamma, dbeta);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_affine_cols(const float* X, int M, int C, const float* scale, const float* shift, int relu, float* Y,
                                void* stream) {
  if (M < 0 || C <= 0 || C % 4 || !X || !Y || !scale || !shift) return MSDE_EINVAL;
  if (M == 0) return 0;
  const long total = (long)M * (C / 4);
  const int blocks = (int)min((total + 255) / 256, (long)msde_num_cus() * 8);
  MSDE_LAUNCH(affine_cols_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), X, M, C, scale, shift, relu, Y);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_bn_bwd_cols(const float* G, int ldg, const float* Z, int ldz, const float* p, const float* w, const float* u,
                                const float* xf3, const float* xf4, int M, const int* m_valid, int C, float* out, int ldo,
                                void* stream) {
  if (M <= 0 || C <= 0) return 0;
  if (C % 4 || ldg % 4 || ldz % 4 || ldo % 4 || !G || !Z || !p || !w || !u || !out || ((xf3 == nullptr) != (xf4 == nullptr)))
    return MSDE_EINVAL;
  const long total = (long)M * (C >> 2);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  MSDE_LAUNCH(bn_bwd_cols_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), G, ldg, Z, ldz, p, w, u, xf3, xf4, M, m_valid, C,
              out, ldo);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_bn_bwd_fin_cols(const float* stats, int strips, const float* gamma, const float* mean, const float* rstd,
                                    const float* G, int ldg, const float* Z, int ldz, const float* xf3, const float* xf4, int M,
                                    const int* m_valid, int C, float* out, int ldo, float* dgamma, float* dbeta, void* stream) {
  if (M <= 0 || C <= 0) return 0;
  if (C % 4 || ldg % 4 || ldz % 4 || ldo % 4 || !stats || strips <= 0 || !mean || !rstd || !G || !Z || !out ||
      ((xf3 == nullptr) != (xf4 == nullptr)))
    return MSDE_EINVAL;
  const int cb = (C + 63) / 64;
  // row chunk: ~one resident round of workgroups per CU or more (C = 600: 10 column blocks x 128-row chunks; C = 300: 5 x 64)
  const int rows_per_wg = (long)cb * ((M + 127) / 128) >= msde_num_cus() ? 128 : 64;
  dim3 grid(cb, (M + rows_per_wg - 1) / rows_per_wg);
  MSDE_LAUNCH(bn_bwd_fin_cols_kernel, grid, dim3(256), 0, as_stream(stream), stats, strips, gamma, mean, rstd, G, ldg, Z, ldz,
              xf3, xf4, M, m_valid, C, rows_per_wg, out, ldo, dgamma, dbeta);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_bn_bwd_colstats(const float* G, const float* Z, const float* Y, const float* mean, int M,
                                    const int* m_valid, int C, float* stats, void* stream) {
  if (M <= 0 || C <= 0 || C % 4 || !G || !Z || !mean || !stats) return MSDE_EINVAL;
  MSDE_LAUNCH(bn_bwd_colstats_kernel, dim3((C + 63) / 64, (M + 63) / 64), dim3(256), 0, as_stream(stream), G, Z, Y, mean, M,
              m_valid, C, stats);
  MSDE_CHECK_LAUNCH();
  return 0;
}
