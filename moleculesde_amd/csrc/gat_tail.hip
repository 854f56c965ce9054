// gat_tail.hip — everything of a GATLayer after the attention, per node and in one kernel each way
// (equivariant_scorenetwork.py:27-38,142):
//     y1  = res + LayerNorm1(x)                    x = attention output, res = the layer's input
//     a   = Dropout_p(SiLU(W0 y1 + b0))            FFN[0..2]
//     x2  = W3 a + b3                              FFN[3]
//     out = y1 + LayerNorm2(x2)                    [-> SiLU(out) between the two convolutions of a block]
// At hidden size 32 each of these six stages is a launch-bound 5 us kernel on 3.6 k rows; fused, four lanes own
// one row (8 columns each, the row all-gathered by shuffles for the two 32 x 32 products), the weight matrices sit
// in LDS in the layout that makes the four lanes' reads conflict free,
// and the only HBM traffic is the row in / row out plus the three saved intermediates (y1, h0 = W0 y1 + b0, x2).
// Backward: one kernel recomputes the normalisations / activations from the saved rows and produces both
// input gradients, the two (gradient, input) pairs the weight-gradient GEMMs need, and per-workgroup partial
// sums of the four LayerNorm parameter gradients (summed later in a fixed order).
#include "msde_common.h"

#define GT_THREADS 256
#ifndef GT_LPR
#define GT_LPR 8                       // lanes per row (4: 58 workgroups at N = 3588, 232 waves for 1024 SIMDs; 8: twice that)
#endif
#define GT_ROWS (GT_THREADS / GT_LPR)  // rows per workgroup

__device__ __forceinline__ float gt_sigmoid(float x) {
  float e = expf(-fabsf(x));
  float r = 1.f / (1.f + e);
  return x >= 0.f ? r : e * r;
}

// Four adjacent lanes own one row: lane q holds the C = D/4 columns [q*C, q*C + C).  A row of D = 32 floats is one
// 128-byte line read by its four lanes.
template <int C>
__device__ __forceinline__ float gt_row_sum(float s) {
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  if (GT_LPR >= 8) s += __shfl_xor(s, 4);
  if (GT_LPR >= 16) s += __shfl_xor(s, 8);
  return s;
}
template <int C>
__device__ __forceinline__ void gt_layernorm(const float (&v)[C], float eps, float& mu, float& rs) {
  constexpr float invD = 1.f / (float)(C * GT_LPR);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < C; ++k) s += v[k];
  mu = gt_row_sum<C>(s) * invD;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < C; ++k) q = fmaf(v[k] - mu, v[k] - mu, q);
  rs = rsqrtf(gt_row_sum<C>(q) * invD + eps);
}
// all-gather of the row across its four lanes: full[r*C + c] = v[c] of lane r
template <int C>
__device__ __forceinline__ void gt_gather(const float (&v)[C], float (&full)[C * GT_LPR]) {
  const int base = (threadIdx.x & 63) & ~(GT_LPR - 1);
#pragma unroll
  for (int r = 0; r < GT_LPR; ++r)
#pragma unroll
    for (int c = 0; c < C; ++c) full[r * C + c] = __shfl(v[c], base + r);
}
// out[jj] = bias[q*C + jj] + sum_k W[q*C + jj][k] full[k]; Wp is the PERMUTED image [jj][q][k] so that the four lanes
// of a row read four consecutive 128-byte rows (no bank conflicts)
template <int C>
__device__ __forceinline__ void gt_matvec(const float* __restrict__ Wp, const float* __restrict__ bs, int q,
                                          const float (&full)[C * GT_LPR], float (&out)[C]) {
  constexpr int D = C * GT_LPR;
#pragma unroll
  for (int jj = 0; jj < C; ++jj) {
    float acc = bs[q * C + jj];
    const float* wr = Wp + (jj * GT_LPR + q) * D;
#pragma unroll
    for (int k = 0; k < D; k += 4) {
      float4 w = *reinterpret_cast<const float4*>(wr + k);
      acc = fmaf(w.x, full[k], acc); acc = fmaf(w.y, full[k + 1], acc);
      acc = fmaf(w.z, full[k + 2], acc); acc = fmaf(w.w, full[k + 3], acc);
    }
    out[jj] = acc;
  }
}
// out[kk] += sum_j W[j][q*C + kk] gfull[j]   (natural [j][k] image: the four lanes read one contiguous row)
template <int C>
__device__ __forceinline__ void gt_matvec_t(const float* __restrict__ Ws, int q, const float (&gfull)[C * GT_LPR],
                                            float (&out)[C]) {
  constexpr int D = C * GT_LPR;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float gj = gfull[j];
#pragma unroll
    for (int kk = 0; kk < C; kk += 4) {
      float4 w = *reinterpret_cast<const float4*>(Ws + j * D + q * C + kk);
      out[kk] = fmaf(w.x, gj, out[kk]); out[kk + 1] = fmaf(w.y, gj, out[kk + 1]);
      out[kk + 2] = fmaf(w.z, gj, out[kk + 2]); out[kk + 3] = fmaf(w.w, gj, out[kk + 3]);
    }
  }
}
template <int C>
__device__ __forceinline__ void gt_load(const float* __restrict__ p, size_t off, float (&v)[C]) {
#pragma unroll
  for (int k = 0; k < C; k += 4) {
    float4 t = *reinterpret_cast<const float4*>(p + off + k);
    v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
  }
}
template <int C>
__device__ __forceinline__ void gt_store(float* __restrict__ p, size_t off, const float (&v)[C]) {
#pragma unroll
  for (int k = 0; k < C; k += 4) *reinterpret_cast<float4*>(p + off + k) = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
}

template <int D>
__global__ void __launch_bounds__(GT_THREADS)
gat_tail_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ ln1_g,
                    const float* __restrict__ ln1_b, const float* __restrict__ W0, const float* __restrict__ b0,
                    const float* __restrict__ W3, const float* __restrict__ b3, const float* __restrict__ ln2_g,
                    const float* __restrict__ ln2_b, int N, float eps1, float eps2, float p_drop, unsigned long long seed,
                    const unsigned long long* __restrict__ seed_dev, int silu_out, float* __restrict__ out,
                    float* __restrict__ y1_o, float* __restrict__ h0_o, float* __restrict__ x2_o) {
  constexpr int C = D / GT_LPR;
  __shared__ __attribute__((aligned(16))) float W0p[D * D], W3p[D * D];   // permuted images [jj][q][k]
  __shared__ float prm[6 * D];          // ln1_g, ln1_b, b0, b3, ln2_g, ln2_b
  for (int t = threadIdx.x; t < D * D; t += GT_THREADS) {
    int j = t / D, k = t % D;
    int dst = ((j % C) * GT_LPR + j / C) * D + k;
    W0p[dst] = W0[t]; W3p[dst] = W3[t];
  }
  for (int t = threadIdx.x; t < D; t += GT_THREADS) {
    prm[t] = ln1_g[t]; prm[D + t] = ln1_b[t]; prm[2 * D + t] = b0[t]; prm[3 * D + t] = b3[t];
    prm[4 * D + t] = ln2_g[t]; prm[5 * D + t] = ln2_b[t];
  }
  __syncthreads();
  if (seed_dev) seed += seed_dev[0] * 0x100000001B3ull;
  const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const int q = threadIdx.x & (GT_LPR - 1);
  size_t i = (size_t)blockIdx.x * GT_ROWS + threadIdx.x / GT_LPR;
  const bool live = i < (size_t)N;
  if (!live) i = (size_t)N - 1;                    // keep the quad's shuffles well defined; nothing is stored
  const size_t off = i * D + q * C;
  float v[C], y1[C], h[C], full[D];
  gt_load<C>(x, off, v);
  gt_load<C>(res, off, y1);
  float mu, rs;
  gt_layernorm<C>(v, eps1, mu, rs);
#pragma unroll
  for (int k = 0; k < C; ++k) y1[k] += fmaf((v[k] - mu) * rs, prm[q * C + k], prm[D + q * C + k]);
  gt_gather<C>(y1, full);
  gt_matvec<C>(W0p, prm + 2 * D, q, full, h);
  if (live) { gt_store<C>(y1_o, off, y1); gt_store<C>(h0_o, off, h); }
#pragma unroll
  for (int j = 0; j < C; ++j) {
    float s = h[j] * gt_sigmoid(h[j]);
    if (p_drop > 0.f) s = msde_uniform(seed, (unsigned long long)(off + j)) >= p_drop ? s * scale : 0.f;
    h[j] = s;
  }
  gt_gather<C>(h, full);
  gt_matvec<C>(W3p, prm + 3 * D, q, full, v);      // v = x2
  if (live) gt_store<C>(x2_o, off, v);
  gt_layernorm<C>(v, eps2, mu, rs);
#pragma unroll
  for (int k = 0; k < C; ++k) {
    float o = y1[k] + fmaf((v[k] - mu) * rs, prm[4 * D + q * C + k], prm[5 * D + q * C + k]);
    v[k] = silu_out ? o * gt_sigmoid(o) : o;
  }
  if (live) gt_store<C>(out, off, v);
}

template <int D>
__global__ void __launch_bounds__(GT_THREADS)
gat_tail_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ x, const float* __restrict__ y1_s,
                    const float* __restrict__ h0_s, const float* __restrict__ x2_s, const float* __restrict__ ln1_g,
                    const float* __restrict__ W0, const float* __restrict__ W3, const float* __restrict__ ln2_g,
                    const float* __restrict__ ln2_b, int N, const int* __restrict__ Ndev, float eps1, float eps2,
                    float p_drop, unsigned long long seed, const unsigned long long* __restrict__ seed_dev, int silu_out,
                    float* __restrict__ g_x, float* __restrict__ g_res, float* __restrict__ g_x2_o,
                    float* __restrict__ a_o, float* __restrict__ g_h0_o, float* __restrict__ ln_part) {
  constexpr int C = D / GT_LPR;
  __shared__ __attribute__((aligned(16))) float W0s[D * D], W3s[D * D];   // natural images [j][k]
  __shared__ float prm[3 * D];                 // ln1_g, ln2_g, ln2_b
  __shared__ float red[GT_ROWS][4 * D + 1];    // per-row LayerNorm parameter contributions (odd stride)
  for (int t = threadIdx.x; t < D * D; t += GT_THREADS) { W0s[t] = W0[t]; W3s[t] = W3[t]; }
  for (int t = threadIdx.x; t < D; t += GT_THREADS) { prm[t] = ln1_g[t]; prm[D + t] = ln2_g[t]; prm[2 * D + t] = ln2_b[t]; }
  __syncthreads();
  if (seed_dev) seed += seed_dev[0] * 0x100000001B3ull;
  const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const int q = threadIdx.x & (GT_LPR - 1), rowl = threadIdx.x / GT_LPR;
  size_t i = (size_t)blockIdx.x * GT_ROWS + rowl;
  const bool live = i < (size_t)N;
  // rows past the row bound are still computed and stored (finite, never stale) but add nothing to the parameter sums
  const float lv = i < (size_t)msde_true_rows(N, Ndev) ? 1.f : 0.f;
  if (!live) i = (size_t)N - 1;
  const size_t off = i * D + q * C;
  float g[C], u[C], w[C], t2[C], full[D];
  gt_load<C>(g_out, off, g);
  gt_load<C>(x2_s, off, u);                   // u = x2
  gt_load<C>(y1_s, off, w);                   // w = y1
  float mu, rs;
  gt_layernorm<C>(u, eps2, mu, rs);
#pragma unroll
  for (int k = 0; k < C; ++k) u[k] = (u[k] - mu) * rs;          // u = xhat2
  if (silu_out) {
#pragma unroll
    for (int k = 0; k < C; ++k) {
      float o = w[k] + fmaf(u[k], prm[D + q * C + k], prm[2 * D + q * C + k]);
      float sg = gt_sigmoid(o);
      g[k] *= sg * (1.f + o * (1.f - sg));
    }
  }
  // out = y1 + LN2(x2): gradient g goes to y1 unchanged and through the normalisation to x2
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    red[rowl][q * C + k] = lv * g[k] * u[k];           // d ln2_g
    red[rowl][D + q * C + k] = lv * g[k];              // d ln2_b
    float gg = g[k] * prm[D + q * C + k];
    c1 += gg; c2 = fmaf(gg, u[k], c2);
  }
  c1 = gt_row_sum<C>(c1) * (1.f / D); c2 = gt_row_sum<C>(c2) * (1.f / D);
#pragma unroll
  for (int k = 0; k < C; ++k) t2[k] = rs * (g[k] * prm[D + q * C + k] - c1 - u[k] * c2);   // t2 = g_x2
  if (live) gt_store<C>(g_x2_o, off, t2);
  // x2 = W3 a + b3
  gt_gather<C>(t2, full);
#pragma unroll
  for (int k = 0; k < C; ++k) u[k] = 0.f;
  gt_matvec_t<C>(W3s, q, full, u);            // u = g_a
  gt_load<C>(h0_s, off, t2);                  // t2 = h0
#pragma unroll
  for (int j = 0; j < C; ++j) {
    float sg = gt_sigmoid(t2[j]);
    float a = t2[j] * sg, da = sg * (1.f + t2[j] * (1.f - sg));
    float m = 1.f;
    if (p_drop > 0.f) m = msde_uniform(seed, (unsigned long long)(off + j)) >= p_drop ? scale : 0.f;
    t2[j] = a * m;                            // a (input of the W3 weight gradient)
    u[j] = u[j] * m * da;                     // g_h0
  }
  if (live) { gt_store<C>(a_o, off, t2); gt_store<C>(g_h0_o, off, u); }
  // h0 = W0 y1 + b0: g_y1 = g + W0^T g_h0
  gt_gather<C>(u, full);
  gt_matvec_t<C>(W0s, q, full, g);            // g = g_y1
  if (live) gt_store<C>(g_res, off, g);
  // y1 = res + LN1(x)
  gt_load<C>(x, off, u);
  gt_layernorm<C>(u, eps1, mu, rs);
  c1 = 0.f; c2 = 0.f;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    u[k] = (u[k] - mu) * rs;                  // xhat1
    red[rowl][2 * D + q * C + k] = lv * g[k] * u[k];   // d ln1_g
    red[rowl][3 * D + q * C + k] = lv * g[k];          // d ln1_b
    float gg = g[k] * prm[q * C + k];
    c1 += gg; c2 = fmaf(gg, u[k], c2);
  }
  c1 = gt_row_sum<C>(c1) * (1.f / D); c2 = gt_row_sum<C>(c2) * (1.f / D);
#pragma unroll
  for (int k = 0; k < C; ++k) t2[k] = rs * (g[k] * prm[q * C + k] - c1 - u[k] * c2);
  if (live) gt_store<C>(g_x, off, t2);
  __syncthreads();
  // per-workgroup partial sums of [d ln2_g | d ln2_b | d ln1_g | d ln1_b], rows added in row order
  for (int t = threadIdx.x; t < 4 * D; t += GT_THREADS) {
    float acc = 0.f;
#pragma unroll 8
    for (int r = 0; r < GT_ROWS; ++r) acc += red[r][t];
    ln_part[(size_t)blockIdx.x * 4 * D + t] = acc;
  }
}

extern "C" int msde_gat_tail_blocks(int N) { return (N + GT_ROWS - 1) / GT_ROWS; }

extern "C" int msde_gat_tail_fwd(const float* x, const float* res, const float* ln1_g, const float* ln1_b, const float* W0,
                                 const float* b0, const float* W3, const float* b3, const float* ln2_g,
                                 const float* ln2_b, int N, int D, float eps1, float eps2, float p_drop,
                                 unsigned long long seed, const unsigned long long* seed_dev, int silu_out, float* out,
                                 float* y1, float* h0, float* x2, void* stream) {
  if (N < 0 || !x || !res || !ln1_g || !ln1_b || !W0 || !b0 || !W3 || !b3 || !ln2_g || !ln2_b || !out || !y1 || !h0 || !x2)
    return MSDE_EINVAL;
  if (p_drop < 0.f || p_drop >= 1.f) return MSDE_EINVAL;
  if (D != 32) return MSDE_EUNSUP;
  if (N == 0) return 0;
  MSDE_LAUNCH(gat_tail_fwd_kernel<32>, dim3(msde_gat_tail_blocks(N)), dim3(GT_THREADS), 0, as_stream(stream), x, res, ln1_g,
              ln1_b, W0, b0, W3, b3, ln2_g, ln2_b, N, eps1, eps2, p_drop, seed, seed_dev, silu_out, out, y1, h0, x2);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_gat_tail_bwd(const float* g_out, const float* x, const float* y1, const float* h0, const float* x2,
                                 const float* ln1_g, const float* W0, const float* W3, const float* ln2_g,
                                 const float* ln2_b, int N, int D, float eps1, float eps2, float p_drop,
                                 unsigned long long seed, const unsigned long long* seed_dev, int silu_out, float* g_x,
                                 float* g_res, float* g_x2, float* a, float* g_h0, float* ln_part, const int* rows_dev,
                                 void* stream) {
  if (N < 0 || !g_out || !x || !y1 || !h0 || !x2 || !ln1_g || !W0 || !W3 || !ln2_g || !ln2_b || !g_x || !g_res || !g_x2 ||
      !a || !g_h0 || !ln_part)
    return MSDE_EINVAL;
  if (p_drop < 0.f || p_drop >= 1.f) return MSDE_EINVAL;
  if (D != 32) return MSDE_EUNSUP;
  if (N == 0) return 0;
  MSDE_LAUNCH(gat_tail_bwd_kernel<32>, dim3(msde_gat_tail_blocks(N)), dim3(GT_THREADS), 0, as_stream(stream), g_out, x, y1,
              h0, x2, ln1_g, W0, W3, ln2_g, ln2_b, N, rows_dev, eps1, eps2, p_drop, seed, seed_dev, silu_out, g_x, g_res, g_x2,
              a, g_h0,
              ln_part);
  MSDE_CHECK_LAUNCH();
  return 0;
}
