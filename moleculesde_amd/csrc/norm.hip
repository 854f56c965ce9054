// norm.hip — training-mode BatchNorm1d over rows (atoms or edges) with optional fused ReLU.
// molecule_gnn_model.py:17,159,176-182 (10 BatchNorms of the GIN stack) and
// SDE_model_2D_to_3D.py:265 (BatchNorm over ~35 k edges inside edge_2D_emb).
//
// Forward  = 2 launches: per-(row split, column) shifted-sum partials, then combine (fixed order, Chan's
//            formula) + normalise + affine (+ ReLU) + running-stat update.
// Backward = 2 launches: per-split column sums of dz and dz*xhat, then combine + input gradient.
// Threads hold float4 column groups (16 per 64-column block) by 16 row lanes; 64-row splits put 300-600
// workgroups in flight for the 3.6 k-atom batch, with four 16-B loads per thread outstanding.
#include "msde_common.h"

#define BN_COLS 64   // columns per block
// Thread layout, templated on the vector width V (4 when C % 4 == 0, else 1): CG = 64 / V column groups by
// RL = 256 / CG row lanes (16 x 16 for float4, 64 x 4 scalar).  A wave covers 4 (or 1) rows of 256 B each.
template <int V> struct BnGeo {
  static constexpr int CG = BN_COLS / V, RL = 256 / CG;
};
#define BN_MAXRL 16
#define BN_MAXSPLIT 64   // bn_geometry never makes more row splits than this

template <int V> __device__ __forceinline__ typename VecT<V>::type bn_ld(const float* p) {
  return *reinterpret_cast<const typename VecT<V>::type*>(p);
}
template <int V> __device__ __forceinline__ void bn_st(float* p, typename VecT<V>::type v) {
  *reinterpret_cast<typename VecT<V>::type*>(p) = v;
}
__device__ __forceinline__ float& vref(float4& v, int k) { return reinterpret_cast<float*>(&v)[k]; }
__device__ __forceinline__ float& vref(float& v, int) { return v; }
__device__ __forceinline__ float vget(const float4& v, int k) { return reinterpret_cast<const float*>(&v)[k]; }
__device__ __forceinline__ float vget(const float& v, int) { return v; }
__device__ __forceinline__ float4 vsub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float vsub(float a, float b) { return a - b; }

// merge (nb, mb, m2b) into (n, mean, m2): Chan's parallel-variance update
__device__ __forceinline__ void bn_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
  if (nb > 0.f) {
    float nn = n + nb, d = mb - mean;
    mean += d * (nb / nn);
    m2 += m2b + d * d * (n * nb / nn);
    n = nn;
  }
}

// Per-(row split, column) partials (n, mean, M2).  Shifted sums s1 = sum(x - x0), s2 = sum((x - x0)^2) with
// x0 = the split's first row (close to the mean: no catastrophic cancellation); every row lane of the block
// uses the SAME shift, so lane partials add directly, in lane order.
template <int V>
__global__ void __launch_bounds__(256)
bn_stats_partial_kernel(const float* __restrict__ X, int M, const int* __restrict__ Mdev, int C, int rows_per_split,
                        float* __restrict__ ws) {
  M = msde_true_rows(M, Mdev);          // statistics over the valid rows only (the apply pass still covers all rows)
  using T = typename VecT<V>::type;
  constexpr int CG = BnGeo<V>::CG, RL = BnGeo<V>::RL;
  __shared__ float s_1[BN_MAXRL][BN_COLS], s_2[BN_MAXRL][BN_COLS];
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int c = blockIdx.x * BN_COLS + tx * V;
  const int r0 = blockIdx.y * rows_per_split, r1 = max(min(r0 + rows_per_split, M), r0);
  T a1 = vzero<V>(), a2 = vzero<V>(), b1 = vzero<V>(), b2 = vzero<V>();
  T x0 = vzero<V>();
  if (c < C) {
    x0 = bn_ld<V>(X + (size_t)r0 * C + c);
    int r = r0 + ty;
    for (; r + 3 * RL < r1; r += 4 * RL) {           // four independent rows in flight
      T xa = vsub(bn_ld<V>(X + (size_t)r * C + c), x0), xb = vsub(bn_ld<V>(X + (size_t)(r + RL) * C + c), x0);
      T xc = vsub(bn_ld<V>(X + (size_t)(r + 2 * RL) * C + c), x0), xd = vsub(bn_ld<V>(X + (size_t)(r + 3 * RL) * C + c), x0);
      a1 = vadd(a1, xa); a2 = vfma(xa, xa, a2);
      b1 = vadd(b1, xb); b2 = vfma(xb, xb, b2);
      a1 = vadd(a1, xc); a2 = vfma(xc, xc, a2);
      b1 = vadd(b1, xd); b2 = vfma(xd, xd, b2);
    }
    for (; r < r1; r += RL) {
      T xa = vsub(bn_ld<V>(X + (size_t)r * C + c), x0);
      a1 = vadd(a1, xa); a2 = vfma(xa, xa, a2);
    }
    a1 = vadd(a1, b1); a2 = vadd(a2, b2);
  }
#pragma unroll
  for (int k = 0; k < V; ++k) { s_1[ty][tx * V + k] = vget(a1, k); s_2[ty][tx * V + k] = vget(a2, k); }
  __syncthreads();
  if (threadIdx.x < BN_COLS) {
    const int cc = blockIdx.x * BN_COLS + threadIdx.x;
    if (cc < C) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int l = 0; l < RL; ++l) { s1 += s_1[l][threadIdx.x]; s2 += s_2[l][threadIdx.x]; }
      const float n = (float)(r1 - r0);
      const float xs = X[(size_t)r0 * C + cc];
      float* o = ws + ((size_t)blockIdx.y * C + cc) * 3;
      if (n > 0.f) { o[0] = n; o[1] = xs + s1 / n; o[2] = fmaxf(s2 - s1 * s1 / n, 0.f); }
      else { o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; }          // a split entirely past the valid rows
    }
  }
}

template <int V>
__global__ void __launch_bounds__(256)
bn_fwd_apply_kernel(const float* __restrict__ X, const float* __restrict__ ws, int M, int C, int splits,
                    int rows_per_block, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                    float momentum, float* __restrict__ running_mean, float* __restrict__ running_var, int relu,
                    float* __restrict__ Y, float* __restrict__ save_mean, float* __restrict__ save_rstd) {
  using T = typename VecT<V>::type;
  constexpr int CG = BnGeo<V>::CG, RL = BnGeo<V>::RL;
  __shared__ float s_scale[BN_COLS], s_shift[BN_COLS];
  __shared__ float s_n[4][BN_COLS], s_mean[4][BN_COLS], s_m2[4][BN_COLS];
  {  // 4 lanes per column merge the splits s = l, l+4, ...; the four partials are then merged in lane order
    const int cx = threadIdx.x & (BN_COLS - 1), l = threadIdx.x / BN_COLS;
    const int cc = blockIdx.x * BN_COLS + cx;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (cc < C) {
      // all of this lane's partials are requested first (independent loads), then merged in split order
      float pn[BN_MAXSPLIT / 4], pm[BN_MAXSPLIT / 4], pq[BN_MAXSPLIT / 4];
#pragma unroll
      for (int k = 0; k < BN_MAXSPLIT / 4; ++k) {
        int sidx = l + 4 * k;
        bool ok = sidx < splits;
        const float* p = ws + ((size_t)(ok ? sidx : 0) * C + cc) * 3;
        pn[k] = ok ? p[0] : 0.f; pm[k] = p[1]; pq[k] = p[2];
      }
#pragma unroll
      for (int k = 0; k < BN_MAXSPLIT / 4; ++k) bn_merge(n, mean, m2, pn[k], pm[k], pq[k]);
    }
    s_n[l][cx] = n; s_mean[l][cx] = mean; s_m2[l][cx] = m2;
    __syncthreads();
    if (l == 0 && cc < C) {
#pragma unroll
      for (int k = 1; k < 4; ++k) bn_merge(n, mean, m2, s_n[k][cx], s_mean[k][cx], s_m2[k][cx]);
      float var = m2 / n;
      float rstd = rsqrtf(var + eps);
      float g = gamma ? gamma[cc] : 1.f, b = beta ? beta[cc] : 0.f;
      s_scale[cx] = g * rstd;
      s_shift[cx] = b - mean * g * rstd;
      if (blockIdx.y == 0) {
        save_mean[cc] = mean;
        save_rstd[cc] = rstd;
        if (running_mean) {
          float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
          running_mean[cc] = (1.f - momentum) * running_mean[cc] + momentum * mean;
          running_var[cc] = (1.f - momentum) * running_var[cc] + momentum * unbiased;
        }
      }
    }
    __syncthreads();
  }
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int c = blockIdx.x * BN_COLS + tx * V;
  if (c >= C) return;
  T sc, sh;
#pragma unroll
  for (int k = 0; k < V; ++k) { vref(sc, k) = s_scale[tx * V + k]; vref(sh, k) = s_shift[tx * V + k]; }
  const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, M);
  int r = r0 + ty;
  for (; r + 3 * RL < r1; r += 4 * RL) {
    T xa = bn_ld<V>(X + (size_t)r * C + c), xb = bn_ld<V>(X + (size_t)(r + RL) * C + c);
    T xc = bn_ld<V>(X + (size_t)(r + 2 * RL) * C + c), xd = bn_ld<V>(X + (size_t)(r + 3 * RL) * C + c);
    xa = vfma(xa, sc, sh); xb = vfma(xb, sc, sh); xc = vfma(xc, sc, sh); xd = vfma(xd, sc, sh);
    if (relu) { xa = vrelu(xa); xb = vrelu(xb); xc = vrelu(xc); xd = vrelu(xd); }
    bn_st<V>(Y + (size_t)r * C + c, xa); bn_st<V>(Y + (size_t)(r + RL) * C + c, xb);
    bn_st<V>(Y + (size_t)(r + 2 * RL) * C + c, xc); bn_st<V>(Y + (size_t)(r + 3 * RL) * C + c, xd);
  }
  for (; r < r1; r += RL) {
    T y = vfma(bn_ld<V>(X + (size_t)r * C + c), sc, sh);
    if (relu) y = vrelu(y);
    bn_st<V>(Y + (size_t)r * C + c, y);
  }
}

// dz = dY gated by the fused ReLU (y = gamma*xhat + beta > 0), xhat = (x - mean) * rstd
template <int V>
__device__ __forceinline__ void bn_dz_xhat(typename VecT<V>::type x, typename VecT<V>::type& d, typename VecT<V>::type mu,
                                           typename VecT<V>::type rs, typename VecT<V>::type g,
                                           typename VecT<V>::type b, int relu, typename VecT<V>::type& xh) {
#pragma unroll
  for (int k = 0; k < V; ++k) {
    float h = (vget(x, k) - vget(mu, k)) * vget(rs, k);
    vref(xh, k) = h;
    if (relu && !(fmaf(h, vget(g, k), vget(b, k)) > 0.f)) vref(d, k) = 0.f;
  }
}

template <int V>
__device__ __forceinline__ void bn_col_params(const float* mean, const float* rstd, const float* gamma, const float* beta,
                                              int c, typename VecT<V>::type& mu, typename VecT<V>::type& rs,
                                              typename VecT<V>::type& g, typename VecT<V>::type& b) {
#pragma unroll
  for (int k = 0; k < V; ++k) {
    vref(mu, k) = mean[c + k]; vref(rs, k) = rstd[c + k];
    vref(g, k) = gamma ? gamma[c + k] : 1.f; vref(b, k) = beta ? beta[c + k] : 0.f;
  }
}

// partial column sums of dz and dz*xhat
template <int V>
__global__ void __launch_bounds__(256)
bn_bwd_partial_kernel(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ mean,
                      const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                      int relu, int M, const int* __restrict__ Mdev, int C, int rows_per_split, float* __restrict__ ws) {
  using T = typename VecT<V>::type;
  constexpr int CG = BnGeo<V>::CG, RL = BnGeo<V>::RL;
  __shared__ float s_a[BN_MAXRL][BN_COLS], s_b[BN_MAXRL][BN_COLS];
  M = msde_true_rows(M, Mdev);
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int c = blockIdx.x * BN_COLS + tx * V;
  const int r0 = blockIdx.y * rows_per_split, r1 = min(r0 + rows_per_split, M);
  T sa = vzero<V>(), sb = vzero<V>(), sa2 = vzero<V>(), sb2 = vzero<V>();
  if (c < C) {
    T mu, rs, g, b;
    bn_col_params<V>(mean, rstd, gamma, beta, c, mu, rs, g, b);
    int r = r0 + ty;
    for (; r + 3 * RL < r1; r += 4 * RL) {       // four independent rows in flight
      T x0 = bn_ld<V>(X + (size_t)r * C + c), x1 = bn_ld<V>(X + (size_t)(r + RL) * C + c);
      T x2 = bn_ld<V>(X + (size_t)(r + 2 * RL) * C + c), x3 = bn_ld<V>(X + (size_t)(r + 3 * RL) * C + c);
      T d0 = bn_ld<V>(dY + (size_t)r * C + c), d1 = bn_ld<V>(dY + (size_t)(r + RL) * C + c);
      T d2 = bn_ld<V>(dY + (size_t)(r + 2 * RL) * C + c), d3 = bn_ld<V>(dY + (size_t)(r + 3 * RL) * C + c);
      T h0, h1, h2, h3;
      bn_dz_xhat<V>(x0, d0, mu, rs, g, b, relu, h0); bn_dz_xhat<V>(x1, d1, mu, rs, g, b, relu, h1);
      bn_dz_xhat<V>(x2, d2, mu, rs, g, b, relu, h2); bn_dz_xhat<V>(x3, d3, mu, rs, g, b, relu, h3);
      sa = vadd(sa, d0); sb = vfma(d0, h0, sb);
      sa2 = vadd(sa2, d1); sb2 = vfma(d1, h1, sb2);
      sa = vadd(sa, d2); sb = vfma(d2, h2, sb);
      sa2 = vadd(sa2, d3); sb2 = vfma(d3, h3, sb2);
    }
    for (; r < r1; r += RL) {
      T x0 = bn_ld<V>(X + (size_t)r * C + c), d0 = bn_ld<V>(dY + (size_t)r * C + c), h0;
      bn_dz_xhat<V>(x0, d0, mu, rs, g, b, relu, h0);
      sa = vadd(sa, d0); sb = vfma(d0, h0, sb);
    }
    sa = vadd(sa, sa2); sb = vadd(sb, sb2);
  }
#pragma unroll
  for (int k = 0; k < V; ++k) { s_a[ty][tx * V + k] = vget(sa, k); s_b[ty][tx * V + k] = vget(sb, k); }
  __syncthreads();
  if (threadIdx.x < BN_COLS) {
    const int cc = blockIdx.x * BN_COLS + threadIdx.x;
    if (cc < C) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int l = 0; l < RL; ++l) { a += s_a[l][threadIdx.x]; b += s_b[l][threadIdx.x]; }
      float* o = ws + ((size_t)blockIdx.y * C + cc) * 2;
      o[0] = a; o[1] = b;
    }
  }
}

template <int V>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ mean,
                    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                    int relu, const float* __restrict__ ws, int M, const int* __restrict__ Mdev, int C, int splits,
                    int rows_per_block, float* __restrict__ dX, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  using T = typename VecT<V>::type;
  constexpr int CG = BnGeo<V>::CG, RL = BnGeo<V>::RL;
  __shared__ float s_db[BN_COLS], s_dg[BN_COLS];
  __shared__ float s_pa[4][BN_COLS], s_pb[4][BN_COLS];
  {
    const int cx = threadIdx.x & (BN_COLS - 1), l = threadIdx.x / BN_COLS;
    const int cc = blockIdx.x * BN_COLS + cx;
    float pa = 0.f, pb = 0.f;
    if (cc < C) {
      float qa[BN_MAXSPLIT / 4], qb[BN_MAXSPLIT / 4];
#pragma unroll
      for (int k = 0; k < BN_MAXSPLIT / 4; ++k) {
        int sidx = l + 4 * k;
        bool ok = sidx < splits;
        const float* p = ws + ((size_t)(ok ? sidx : 0) * C + cc) * 2;
        qa[k] = ok ? p[0] : 0.f; qb[k] = ok ? p[1] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < BN_MAXSPLIT / 4; ++k) { pa += qa[k]; pb += qb[k]; }
    }
    s_pa[l][cx] = pa; s_pb[l][cx] = pb;
    __syncthreads();
    if (l == 0 && cc < C) {
      float sa = ((s_pa[0][cx] + s_pa[1][cx]) + s_pa[2][cx]) + s_pa[3][cx];
      float sb = ((s_pb[0][cx] + s_pb[1][cx]) + s_pb[2][cx]) + s_pb[3][cx];
      s_db[cx] = sa; s_dg[cx] = sb;
      if (blockIdx.y == 0) {
        if (dbeta) dbeta[cc] = sa;
        if (dgamma) dgamma[cc] = sb;
      }
    }
    __syncthreads();
  }
  const int tx = threadIdx.x % CG, ty = threadIdx.x / CG;
  const int c = blockIdx.x * BN_COLS + tx * V;
  if (c >= C) return;
  T mu, rs, g, b;
  bn_col_params<V>(mean, rstd, gamma, beta, c, mu, rs, g, b);
  const int Mv = msde_true_rows(M, Mdev);
  const float invM = 1.f / (float)Mv;      // mean over the valid rows; dX is written for all rows -- ZERO behind the row bound:
  // a padded row has no loss gradient, but "d - mean(d) - xhat mean(d xhat)" is not zero for d = 0, and what flows on from it
  // reaches reductions over rows further up (GIN's eps / bond-table gradients): found by test_bucket_step_matches_exact_batch
  // with the unfused BatchNorm (2 % gradient error in the deep GIN layers)
  T kb, kg, grs = vmul(g, rs);
#pragma unroll
  for (int k = 0; k < V; ++k) { vref(kb, k) = s_db[tx * V + k] * invM; vref(kg, k) = s_dg[tx * V + k] * invM; }
  const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, M);
  auto one = [&](T x, T d, int row) -> T {
    T h;
    bn_dz_xhat<V>(x, d, mu, rs, g, b, relu, h);
    T o;
    const float live = row < Mv ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) vref(o, k) = live * (vget(grs, k) * (vget(d, k) - vget(kb, k) - vget(h, k) * vget(kg, k)));
    return o;
  };
  int r = r0 + ty;
  for (; r + RL < r1; r += 2 * RL) {
    T x0 = bn_ld<V>(X + (size_t)r * C + c), x1 = bn_ld<V>(X + (size_t)(r + RL) * C + c);
    T d0 = bn_ld<V>(dY + (size_t)r * C + c), d1 = bn_ld<V>(dY + (size_t)(r + RL) * C + c);
    bn_st<V>(dX + (size_t)r * C + c, one(x0, d0, r));
    bn_st<V>(dX + (size_t)(r + RL) * C + c, one(x1, d1, r + RL));
  }
  for (; r < r1; r += RL) bn_st<V>(dX + (size_t)r * C + c, one(bn_ld<V>(X + (size_t)r * C + c), bn_ld<V>(dY + (size_t)r * C + c), r));
}

static inline void bn_geometry(int M, int* splits, int* rows) {
  int s = (M + 63) / 64;            // >= 64 rows per split: (C/64) x splits workgroups fill the chip
  if (s > BN_MAXSPLIT) s = BN_MAXSPLIT;
  if (s < 1) s = 1;
  *rows = (M + s - 1) / s;
  *splits = (M + *rows - 1) / *rows;
}

#define BN_RL 4   // row lanes of the scalar column-sum kernels
// ---- column sums (bias gradients of the library-GEMM weight-gradient path): two launches, fixed order
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float* __restrict__ X, int M, const int* __restrict__ Mdev, int C, int rows_per_split,
                      float* __restrict__ ws) {
  __shared__ float s_a[BN_RL][BN_COLS];
  M = msde_true_rows(M, Mdev);
  const int tx = threadIdx.x & (BN_COLS - 1), ty = threadIdx.x / BN_COLS;
  const int c = blockIdx.x * BN_COLS + tx;
  const int r0 = blockIdx.y * rows_per_split, r1 = min(r0 + rows_per_split, M);
  float sa = 0.f;
  if (c < C)
    for (int r = r0 + ty; r < r1; r += BN_RL) sa += X[(size_t)r * C + c];
  s_a[ty][tx] = sa;
  __syncthreads();
  if (ty == 0 && c < C) {
#pragma unroll
    for (int l = 1; l < BN_RL; ++l) sa += s_a[l][tx];
    ws[(size_t)blockIdx.y * C + c] = sa;
  }
}

__global__ void __launch_bounds__(256)
colsum_final_kernel(const float* __restrict__ ws, int splits, int C, float* __restrict__ out) {
  __shared__ float s_p[BN_RL][BN_COLS];
  const int tx = threadIdx.x & (BN_COLS - 1), ty = threadIdx.x / BN_COLS;
  const int c = blockIdx.x * BN_COLS + tx;
  float acc = 0.f;
  if (c < C)
    for (int s = ty; s < splits; s += BN_RL) acc += ws[(size_t)s * C + c];
  s_p[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c < C) out[c] = ((s_p[0][tx] + s_p[1][tx]) + s_p[2][tx]) + s_p[3][tx];
}

extern "C" int msde_colsum(const float* X, int M, int C, float* out, float* workspace, const int* rows_dev, void* stream) {
  if (M < 0 || C <= 0 || !X || !out || !workspace) return MSDE_EINVAL;
  hipStream_t st = as_stream(stream);
  if (M == 0) return msde_zero_words(out, (size_t)C, st);
  int splits, rows;
  bn_geometry(M, &splits, &rows);
  if (splits == 1) {     // one split: its partial sums are the result (no second launch)
    MSDE_LAUNCH(colsum_partial_kernel, dim3((C + BN_COLS - 1) / BN_COLS, 1), dim3(256), 0, st, X, M, rows_dev, C, rows, out);
    MSDE_CHECK_LAUNCH();
    return 0;
  }
  MSDE_LAUNCH(colsum_partial_kernel, dim3((C + BN_COLS - 1) / BN_COLS, splits), dim3(256), 0, st, X, M, rows_dev, C, rows,
              workspace);
  MSDE_CHECK_LAUNCH();
  MSDE_LAUNCH(colsum_final_kernel, dim3((C + BN_COLS - 1) / BN_COLS), dim3(256), 0, st, (const float*)workspace, splits, C,
              out);
  MSDE_CHECK_LAUNCH();
  return 0;
}

static inline bool bn_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int msde_bn_workspace_floats(int M, int C) {
  int splits, rows;
  bn_geometry(M, &splits, &rows);
  return splits * C * 3;
}

extern "C" int msde_bn_fwd(const float* X, int M, int C, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, int relu, float* Y,
                           float* save_mean, float* save_rstd, float* workspace, const int* rows_dev, void* stream) {
  if (M <= 0 || C <= 0 || !X || !Y || !save_mean || !save_rstd || !workspace) return MSDE_EINVAL;
  int splits, rows;
  bn_geometry(M, &splits, &rows);
  dim3 grid((C + BN_COLS - 1) / BN_COLS, splits);
  const bool vec = (C % 4 == 0) && bn_aligned16(X) && bn_aligned16(Y);
  if (vec) {
    MSDE_LAUNCH(bn_stats_partial_kernel<4>, grid, dim3(256), 0, as_stream(stream), X, M, rows_dev, C, rows, workspace);
    MSDE_CHECK_LAUNCH();
    MSDE_LAUNCH(bn_fwd_apply_kernel<4>, grid, dim3(256), 0, as_stream(stream), X, (const float*)workspace, M, C, splits,
                rows, gamma, beta, eps, momentum, running_mean, running_var, relu, Y, save_mean, save_rstd);
  } else {
    MSDE_LAUNCH(bn_stats_partial_kernel<1>, grid, dim3(256), 0, as_stream(stream), X, M, rows_dev, C, rows, workspace);
    MSDE_CHECK_LAUNCH();
    MSDE_LAUNCH(bn_fwd_apply_kernel<1>, grid, dim3(256), 0, as_stream(stream), X, (const float*)workspace, M, C, splits,
                rows, gamma, beta, eps, momentum, running_mean, running_var, relu, Y, save_mean, save_rstd);
  }
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_bn_bwd(const float* dY, const float* X, const float* save_mean, const float* save_rstd,
                           const float* gamma, const float* beta, int relu, int M, int C, float* dX, float* dgamma,
                           float* dbeta, float* workspace, const int* rows_dev, void* stream) {
  if (M <= 0 || C <= 0 || !dY || !X || !save_mean || !save_rstd || !dX || !workspace) return MSDE_EINVAL;
  int splits, rows;
  bn_geometry(M, &splits, &rows);
  dim3 grid((C + BN_COLS - 1) / BN_COLS, splits);
  const bool vec = (C % 4 == 0) && bn_aligned16(X) && bn_aligned16(dY) && bn_aligned16(dX);
  if (vec) {
    MSDE_LAUNCH(bn_bwd_partial_kernel<4>, grid, dim3(256), 0, as_stream(stream), dY, X, save_mean, save_rstd, gamma, beta,
                relu, M, rows_dev, C, rows, workspace);
    MSDE_CHECK_LAUNCH();
    MSDE_LAUNCH(bn_bwd_apply_kernel<4>, grid, dim3(256), 0, as_stream(stream), dY, X, save_mean, save_rstd, gamma, beta,
                relu, (const float*)workspace, M, rows_dev, C, splits, rows, dX, dgamma, dbeta);
  } else {
    MSDE_LAUNCH(bn_bwd_partial_kernel<1>, grid, dim3(256), 0, as_stream(stream), dY, X, save_mean, save_rstd, gamma, beta,
                relu, M, rows_dev, C, rows, workspace);
    MSDE_CHECK_LAUNCH();
    MSDE_LAUNCH(bn_bwd_apply_kernel<1>, grid, dim3(256), 0, as_stream(stream), dY, X, save_mean, save_rstd, gamma, beta,
                relu, (const float*)workspace, M, rows_dev, C, splits, rows, dX, dgamma, dbeta);
  }
  MSDE_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// y = res + LayerNorm(x)  (GATLayer: node_attr + norm(x), equivariant_scorenetwork.py:36,38), rows of D
// floats; one group of TPR lanes per row, the row lives in registers (D <= 4*TPR*LN_MAXV).
// ------------------------------------------------------------------------------------------------
#define LN_MAXV 4
#define LN_BLOCKS 64

__global__ void __launch_bounds__(256)
res_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                         const float* __restrict__ beta, int N, int cols, int tpr, float eps, float* __restrict__ y,
                         float* __restrict__ mean_o, float* __restrict__ rstd_o) {
  const int gpb = 256 / tpr;
  const int lane = threadIdx.x % tpr;
  const float invD = 1.f / (float)(cols * 4);
  for (int i = blockIdx.x * gpb + threadIdx.x / tpr; i < N; i += gridDim.x * gpb) {
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < LN_MAXV; ++u) {
      int c = lane + u * tpr;
      v[u] = c < cols ? reinterpret_cast<const float4*>(x)[(size_t)i * cols + c] : vzero4();
      s += vhsum(v[u]);
    }
    float mu = group_sum(s, tpr) * invD;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < LN_MAXV; ++u) {
      int c = lane + u * tpr;
      if (c < cols) {
        float4 d = make_float4(v[u].x - mu, v[u].y - mu, v[u].z - mu, v[u].w - mu);
        q += vhsum(vmul(d, d));
      }
    }
    float rs = rsqrtf(group_sum(q, tpr) * invD + eps);
#pragma unroll
    for (int u = 0; u < LN_MAXV; ++u) {
      int c = lane + u * tpr;
      if (c < cols) {
        float4 g = reinterpret_cast<const float4*>(gamma)[c], b = reinterpret_cast<const float4*>(beta)[c];
        float4 o = make_float4(fmaf((v[u].x - mu) * rs, g.x, b.x), fmaf((v[u].y - mu) * rs, g.y, b.y),
                               fmaf((v[u].z - mu) * rs, g.z, b.z), fmaf((v[u].w - mu) * rs, g.w, b.w));
        if (res) o = vadd(o, reinterpret_cast<const float4*>(res)[(size_t)i * cols + c]);
        reinterpret_cast<float4*>(y)[(size_t)i * cols + c] = o;
      }
    }
    if (lane == 0) { mean_o[i] = mu; rstd_o[i] = rs; }
  }
}

__global__ void __launch_bounds__(256)
res_layernorm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ gamma,
                         const float* __restrict__ mean, const float* __restrict__ rstd, int N, int cols, int tpr,
                         float* __restrict__ gx, float* __restrict__ ws) {
  extern __shared__ float lds[];      // [gpb][2 * D] partial gamma/beta sums of the block's groups
  const int gpb = 256 / tpr;
  const int lane = threadIdx.x % tpr, grp = threadIdx.x / tpr;
  const int D = cols * 4;
  const float invD = 1.f / (float)D;
  float4 sg[LN_MAXV], sb[LN_MAXV];
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) { sg[u] = vzero4(); sb[u] = vzero4(); }
  for (int i = blockIdx.x * gpb + grp; i < N; i += gridDim.x * gpb) {
    float mu = mean[i], rs = rstd[i];
    float4 gg[LN_MAXV], xh[LN_MAXV];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int u = 0; u < LN_MAXV; ++u) {
      int c = lane + u * tpr;
      gg[u] = vzero4(); xh[u] = vzero4();
      if (c < cols) {
        float4 gv = reinterpret_cast<const float4*>(g)[(size_t)i * cols + c];
        float4 xv = reinterpret_cast<const float4*>(x)[(size_t)i * cols + c];
        float4 gm = reinterpret_cast<const float4*>(gamma)[c];
        xh[u] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        sb[u] = vadd(sb[u], gv);
        sg[u] = vfma(gv, xh[u], sg[u]);
        gg[u] = vmul(gv, gm);
        c1 += vhsum(gg[u]);
        c2 += vhsum(vmul(gg[u], xh[u]));
      }
    }
    c1 = group_sum(c1, tpr) * invD;
    c2 = group_sum(c2, tpr) * invD;
#pragma unroll
    for (int u = 0; u < LN_MAXV; ++u) {
      int c = lane + u * tpr;
      if (c < cols)
        reinterpret_cast<float4*>(gx)[(size_t)i * cols + c] =
            make_float4(rs * (gg[u].x - c1 - xh[u].x * c2), rs * (gg[u].y - c1 - xh[u].y * c2),
                        rs * (gg[u].z - c1 - xh[u].z * c2), rs * (gg[u].w - c1 - xh[u].w * c2));
    }
  }
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    int c = lane + u * tpr;
    if (c < cols) {
      reinterpret_cast<float4*>(lds + (size_t)grp * 2 * D)[c] = sg[u];
      reinterpret_cast<float4*>(lds + (size_t)grp * 2 * D + D)[c] = sb[u];
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * D; t += 256) {
    float acc = 0.f;
    for (int q = 0; q < gpb; ++q) acc += lds[(size_t)q * 2 * D + t];
    ws[(size_t)blockIdx.x * 2 * D + t] = acc;      // [block][gamma D | beta D]
  }
}

extern "C" int msde_res_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, int N,
                                      int D, float eps, float* y, float* mean, float* rstd, void* stream) {
  if (N < 0 || D <= 0 || !x || !gamma || !beta || !y || !mean || !rstd) return MSDE_EINVAL;
  if (D % 4 != 0) return MSDE_EUNSUP;
  int cols = D / 4, tpr = pick_tpr(cols);
  if (cols > tpr * LN_MAXV) return MSDE_EUNSUP;
  if (N == 0) return 0;
  int gpb = 256 / tpr;
  int blocks = (N + gpb - 1) / gpb;
  if (blocks > 1024) blocks = 1024;
  MSDE_LAUNCH(res_layernorm_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, res, gamma, beta, N, cols, tpr,
              eps, y, mean, rstd);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_res_layernorm_bwd(const float* g, const float* x, const float* gamma, const float* mean,
                                      const float* rstd, int N, int D, float* gx, float* ggamma, float* gbeta,
                                      float* workspace /* LN_BLOCKS * 2 * D floats */, void* stream) {
  if (N <= 0 || D <= 0 || !g || !x || !gamma || !mean || !rstd || !gx || !ggamma || !gbeta || !workspace) return MSDE_EINVAL;
  if (D % 4 != 0) return MSDE_EUNSUP;
  int cols = D / 4, tpr = pick_tpr(cols);
  if (cols > tpr * LN_MAXV) return MSDE_EUNSUP;
  int gpb = 256 / tpr;
  hipStream_t st = as_stream(stream);
  MSDE_LAUNCH(res_layernorm_bwd_kernel, dim3(LN_BLOCKS), dim3(256), (size_t)gpb * 2 * D * sizeof(float), st, g, x, gamma,
              mean, rstd, N, cols, tpr, gx, workspace);
  MSDE_CHECK_LAUNCH();
  // workspace is [LN_BLOCKS][2D]: the column-sum finaliser produces [ggamma | gbeta] when they are contiguous,
  // otherwise two launches on the two halves
  if (gbeta == ggamma + D) {
    MSDE_LAUNCH(colsum_final_kernel, dim3((2 * D + BN_COLS - 1) / BN_COLS), dim3(256), 0, st, (const float*)workspace,
                LN_BLOCKS, 2 * D, ggamma);
    MSDE_CHECK_LAUNCH();
  } else {
    return MSDE_EINVAL;
  }
  return 0;
}
