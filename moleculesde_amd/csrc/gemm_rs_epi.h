// gemm_rs_epi.h — epilogue building blocks shared by the node-level fp32 matrix-core GEMM kernels (gemm_rs.hip: row strips,
// gemm_t2.hip: 2-D tiles): bias, pre-activation store, activation or its derivative, residual / accumulate, and the per-strip
// column statistics of the fused BatchNorm, for one column segment of W interleaved 16 x 16 accumulator tiles.
#pragma once
#include "gemm_rs.h"

// ---- activations (same fast forms as gemm_ex.hip) -----------------------------------------------------------------
__device__ __forceinline__ float rs_act(int act, float z) {
  switch (act) {
    case MSDE_ACT_TANH: return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * z));
    case MSDE_ACT_SILU: return z * __frcp_rn(1.f + __expf(-z));
    case MSDE_ACT_ELU: return z > 0.f ? z : __expf(z) - 1.f;
    case MSDE_ACT_SSP: return (z > 20.f ? z : __logf(1.f + __expf(z))) - 0.6931471805599453f;
    case MSDE_ACT_RELU: return fmaxf(z, 0.f);
    default: return z;
  }
}
__device__ __forceinline__ float rs_dact(int act, float r) {
  switch (act) {
    case MSDE_ACT_TANH: return 1.f - r * r;
    case MSDE_ACT_SILU: { const float s = __frcp_rn(1.f + __expf(-r)); return s * (1.f + r * (1.f - s)); }
    case MSDE_ACT_ELU: return r > 0.f ? 1.f : r + 1.f;
    case MSDE_ACT_SSP: return __frcp_rn(1.f + __expf(-r));
    case MSDE_ACT_RELU: return r > 0.f ? 1.f : 0.f;
    case MSDE_ACT_SSPO: return 1.f - __expf(-(r + 0.6931471805599453f));    // sigmoid(x) from a = softplus(x) - ln 2
    default: return 1.f;
  }
}

// W consecutive floats: one 16-B / 8-B access when the buffer allows it (`vec`), else scalar
template <int W> __device__ __forceinline__ void rs_ldw(const float* __restrict__ p, float (&o)[4], bool vec) {
  if (W == 4 && vec) { const float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
  else if (W == 2 && vec) { const float2 v = *reinterpret_cast<const float2*>(p); o[0] = v.x; o[1] = v.y; }
  else {
#pragma unroll
    for (int t = 0; t < W; ++t) o[t] = p[t];
  }
}
template <int W> __device__ __forceinline__ void rs_stw(float* __restrict__ p, const float (&o)[4], bool vec) {
  if (W == 4 && vec) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  else if (W == 2 && vec) *reinterpret_cast<float2*>(p) = make_float2(o[0], o[1]);
  else {
#pragma unroll
    for (int t = 0; t < W; ++t) p[t] = o[t];
  }
}

// One column segment of the epilogue: W interleaved tiles starting at tile F; the lane's W values of an output row are W
// CONSECUTIVE columns, so bias, saved activations (R), residual, statistics input and the stores all move as vectors.
// C/D map of v_mfma_f32_16x16x4_f32: lane-column i = lane & 15, row = 4 (lane >> 4) + e for element e.  Every uniform
// choice (activation, derivative, residual, accumulate, statistics) is made ONCE, outside the element loops: a chain of
// scalar branches per element cost the plain products a microsecond.
template <int RT, int T, int W, int F>
__device__ __forceinline__ void rs_epi_segment(const msde_rs_desc& d, f32x4 (&acc)[T][RT], int colb, int m0, int strip,
                                               int strip_rows) {
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const int M = d.M, N = d.N;
  if (colb >= N) return;                          // (N % W == 0: a lane's W columns are all inside or all outside)
  const bool vec = (d.flags & MSDE_RS_VEC_STORE) != 0;
  const int row0 = m0 + 4 * g;                    // row of (r, e) = row0 + 16 r + e
  // 1. bias
  if (d.bias) {
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    rs_ldw<W>(d.bias + colb, bv, vec);
#pragma unroll
    for (int t = 0; t < W; ++t)
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[F + t][r][e] += bv[t];
  }
  // 2. pre-activation store
  if (d.Z) {
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = row0 + 16 * r + e;
        if (row < M) {
          float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < W; ++t) v[t] = acc[F + t][r][e];
          rs_stw<W>(d.Z + (size_t)row * d.ldz + colb, v, vec);
        }
      }
  }
  // 3. activation / derivative
  if (d.act != MSDE_ACT_NONE) {
    if (d.epi == MSDE_EPI_DACT) {
#define RS_DACT(ACT_)                                                                                          \
  case ACT_:                                                                                                   \
    _Pragma("unroll") for (int r = 0; r < RT; ++r) _Pragma("unroll") for (int e = 0; e < 4; ++e) {             \
      const int row = row0 + 16 * r + e;                                                                       \
      if (row < M) {                                                                                           \
        float rv[4];                                                                                           \
        rs_ldw<W>(d.R + (size_t)row * d.ldr + colb, rv, vec);                                                  \
        _Pragma("unroll") for (int t = 0; t < W; ++t) acc[F + t][r][e] *= rs_dact(ACT_, rv[t]);                \
      }                                                                                                        \
    }                                                                                                          \
    break;
      switch (d.act) {
        RS_DACT(MSDE_ACT_TANH) RS_DACT(MSDE_ACT_SILU) RS_DACT(MSDE_ACT_ELU) RS_DACT(MSDE_ACT_SSP) RS_DACT(MSDE_ACT_RELU)
        RS_DACT(MSDE_ACT_SSPO)
        default: break;
      }
#undef RS_DACT
    } else {
#define RS_ACT(ACT_)                                                                                           \
  case ACT_:                                                                                                   \
    _Pragma("unroll") for (int t = 0; t < W; ++t) _Pragma("unroll") for (int r = 0; r < RT; ++r)               \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) acc[F + t][r][e] = rs_act(ACT_, acc[F + t][r][e]);       \
    break;
      switch (d.act) {
        RS_ACT(MSDE_ACT_TANH) RS_ACT(MSDE_ACT_SILU) RS_ACT(MSDE_ACT_ELU) RS_ACT(MSDE_ACT_SSP) RS_ACT(MSDE_ACT_RELU)
        default: break;
      }
#undef RS_ACT
    }
  }
  // 4. residual, accumulate, store
  if (!d.Res && !(d.flags & MSDE_GEMM_ACCUMULATE)) {
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = row0 + 16 * r + e;
        if (row < M) {
          float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < W; ++t) v[t] = acc[F + t][r][e];
          rs_stw<W>(d.C + (size_t)row * d.ldc + colb, v, vec);
        }
      }
  } else {
    const bool accum = (d.flags & MSDE_GEMM_ACCUMULATE) != 0;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = row0 + 16 * r + e;
        if (row < M) {
          float v[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < W; ++t) v[t] = acc[F + t][r][e];
          if (d.Res) {
            rs_ldw<W>(d.Res + (size_t)row * d.ldres + colb, q, vec);
#pragma unroll
            for (int t = 0; t < W; ++t) v[t] += q[t];
          }
          float* dst = d.C + (size_t)row * d.ldc + colb;
          if (accum) {
            rs_ldw<W>(dst, q, vec);
#pragma unroll
            for (int t = 0; t < W; ++t) v[t] += q[t];
          }
          rs_stw<W>(dst, v, vec);
#pragma unroll
          for (int t = 0; t < W; ++t) acc[F + t][r][e] = v[t];
        }
      }
  }
  if (!d.stats) return;
  // 5. per-strip column statistics of what was stored, over the VALID rows of the strip
  const int mv = d.m_valid ? min(M, d.m_valid[0]) : M;
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  if (d.stats_mode == MSDE_RS_STATS_BNFWD) {        // mean, then squared deviations from it
    const int cnt = max(0, min(strip_rows, mv - m0));
#pragma unroll
    for (int t = 0; t < W; ++t) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) a += (row0 + 16 * r + e < mv) ? acc[F + t][r][e] : 0.f;
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      const float mean = cnt > 0 ? a / (float)cnt : 0.f;
      float q = 0.f;
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dv = acc[F + t][r][e] - mean;
          q += (row0 + 16 * r + e < mv) ? dv * dv : 0.f;
        }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      s0[t] = mean;
      s1[t] = q;
    }
  } else {                                           // MSDE_RS_STATS_BNBWD: sum g, sum g (z - mean[col])
    float mu[4] = {0.f, 0.f, 0.f, 0.f};
    rs_ldw<W>(d.stats_mean + colb, mu, vec);
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = row0 + 16 * r + e;
        if (row < mv) {
          float zz[4];
          rs_ldw<W>(d.stats_z + (size_t)row * d.ld_sz + colb, zz, vec);
#pragma unroll
          for (int t = 0; t < W; ++t) { s0[t] += acc[F + t][r][e]; s1[t] = fmaf(acc[F + t][r][e], zz[t] - mu[t], s1[t]); }
        }
      }
#pragma unroll
    for (int t = 0; t < W; ++t) {
      s0[t] += __shfl_xor(s0[t], 16, 64);
      s0[t] += __shfl_xor(s0[t], 32, 64);
      s1[t] += __shfl_xor(s1[t], 16, 64);
      s1[t] += __shfl_xor(s1[t], 32, 64);
    }
  }
  if (g == 0) {
    float* __restrict__ out = d.stats + (size_t)strip * 2 * N;
    rs_stw<W>(out + colb, s0, vec);
    rs_stw<W>(out + N + colb, s1, vec);
  }
}
