// dd.hip — the small pointwise / per-edge kernels of the TWICE-differentiable SchNet energy path (MD17 force
// fine-tuning, examples/finetune_MD17.py:47-78: F = -dE/dpos with create_graph=True, then loss(E, F).backward()).
//
// The reference differentiates schnet.py:85-125 twice with autograd over ATen operators.  Here every operator on the
// path positions -> energy is a kernel, and the operator set is CLOSED under differentiation: the backward of each op
// is expressed with ops of the same set (moleculesde_amd/dd.py), so autograd can differentiate the first backward pass
// again without ever leaving the library.  This file holds the members that are not GEMMs (csrc/gemm_ex.hip,
// msde_linear_bwd_w) or edge aggregations (msde_cfconv_aggregate_*):
//   unary derivatives of order 0..2 of the shifted softplus (schnet.py:213-216) and the cosine cutoff (:186), the
//   Gaussian smearing (:205-207) and its d/dd, d2/dd2, reciprocal, products (elementwise, row-broadcast, row dot),
//   sums, per-edge coordinate differences and their scatter, row norms, per-molecule reduce / expand, column sums /
//   row broadcast.  Padded radius-edge slots (src < 0) give zeros at every order.
#include "msde_common.h"

#define DD_GRID(n) dim3((unsigned)(((n) + 255) / 256 > 4096 ? 4096 : ((n) + 255) / 256 < 1 ? 1 : ((n) + 255) / 256))

__device__ __forceinline__ float dd_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// kind 0: shifted softplus, order 0..2; kind 1: cosine cutoff 0.5 (cos(pi d / rc) + 1) for d < rc, 0 beyond, order 0..2
// (p0 = rc); kind 2: reciprocal (order ignored); kind 3: SiLU x sigmoid(x), order 0..2 (painn.py activation);
// kind 4: sqrt(x + p0), order 0..2 (painn.py:104).  mask (optional int array): entries < 0 give 0.
// m1, m2 (optional): y = m1 * m2 * f^(order)(x) -- the chain-rule products of the operator's own backward in the same launch
__global__ void dd_unary_kernel(const float* __restrict__ x, const int* __restrict__ mask, long long n, int kind, int order,
                                float p0, const float* __restrict__ m1, const float* __restrict__ m2, float* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = x[i];
    float r;
    if (kind == 0) {
      const float s = dd_sigmoid(v);
      r = order == 0 ? ((v > 20.f ? v : log1pf(expf(v))) - 0.6931471805599453f) : order == 1 ? s : s * (1.f - s);
    } else if (kind == 1) {
      const float w = 3.14159265358979323846f / p0, a = v * w;
      r = order == 0 ? 0.5f * (cosf(a) + 1.f) : order == 1 ? -0.5f * w * sinf(a) : -0.5f * w * w * cosf(a);
      if (v >= p0) r = 0.f;                      // painn_utils.py:150-154 (SchNet's radius edges never get here)
    } else if (kind == 2) {
      r = 1.f / v;
    } else if (kind == 3) {
      const float s = dd_sigmoid(v), t = 1.f - s;
      r = order == 0 ? v * s : order == 1 ? s * (1.f + v * t) : s * t * (2.f + v * (t - s));
    } else {
      const float q = sqrtf(v + p0);
      r = order == 0 ? q : order == 1 ? 0.5f / q : -0.25f / (q * (v + p0));
    }
    if (m1) r *= m1[i];
    if (m2) r *= m2[i];
    if (mask && mask[i] < 0) r = 0.f;
    y[i] = r;
  }
}

// Gaussian smearing rbf_g(d) = exp(c (d - mu_g)^2) and its first / second derivative in d; rows with src < 0 -> 0
__global__ void dd_rbf_kernel(const float* __restrict__ d, const int* __restrict__ src, const float* __restrict__ mu, int E,
                              int G, float c, int order, float* __restrict__ y) {
  const long long n = (long long)E * G;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int e = (int)(i / G), g = (int)(i - (long long)e * G);
    const float t = d[e] - mu[g], r = expf(c * t * t);
    float v = order == 0 ? r : order == 1 ? 2.f * c * t * r : (2.f * c + 4.f * c * c * t * t) * r;
    if (src && src[e] < 0) v = 0.f;
    y[i] = v;
  }
}

// op 0: y = alpha a b; op 1: y = a + b; op 2: y = alpha a
__global__ void dd_binary_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, int op, float alpha,
                                 float* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    y[i] = op == 0 ? alpha * a[i] * b[i] : op == 1 ? a[i] + b[i] : alpha * a[i];
}

// y = x_0 + x_1 + ... + x_{n-1} (n <= 8, summed in index order): the gradient of a tensor with n consumers in ONE launch
struct dd_ptrs8 { const float* p[8]; };
__global__ void dd_sum_n_kernel(dd_ptrs8 s, int n, long long count, float* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    float v = s.p[0][i];
#pragma unroll
    for (int k = 1; k < 8; ++k)
      if (k < n) v += s.p[k][i];
    y[i] = v;
  }
}

// the same for [rows][cols] operands with row strides of their own (a gradient that is a column block of a wider buffer: the edge
// half of the basis MLP's input gradient, equivariant_scorenetwork.py:154-157); cols % 4 == 0, 16-byte aligned rows
struct dd_lds8 { int ld[8]; };
__global__ void dd_sum_rows_n_kernel(dd_ptrs8 s, dd_lds8 l, int n, int rows, int cols4, float* __restrict__ y) {
  const long long count = (long long)rows * cols4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    const long long r = i / cols4;
    const int c = (int)(i - r * cols4) * 4;
    float4 v = *reinterpret_cast<const float4*>(s.p[0] + r * l.ld[0] + c);
#pragma unroll
    for (int k = 1; k < 8; ++k)
      if (k < n) {
        const float4 w = *reinterpret_cast<const float4*>(s.p[k] + r * l.ld[k] + c);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
    *reinterpret_cast<float4*>(y + r * (long long)cols4 * 4 + c) = v;
  }
}

__global__ void dd_mul_rows_kernel(const float* __restrict__ M, const float* __restrict__ s, int E, int K, float* __restrict__ y) {
  const long long n = (long long)E * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = M[i] * s[i / K];
}

// y[e] = sum_k a[e][k] b[e][k]; one wave per row, fixed lane order
__global__ void __launch_bounds__(256) dd_row_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int E, int K,
                                                         float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < E; e += gridDim.x * 4) {
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s = fmaf(a[(size_t)e * K + k], b[(size_t)e * K + k], s);
    s = group_sum(s, 64);
    if (lane == 0) y[e] = s;
  }
}

// diff[e] = pos[src_e] - pos[dst_e]  (0 for padded slots)
__global__ void dd_edge_diff_kernel(const float* __restrict__ pos, const int* __restrict__ src, const int* __restrict__ dst,
                                    int E, float* __restrict__ y) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * E; i += gridDim.x * 256) {
    const int e = i / 3, c = i - 3 * e, s = src[e], t = dst[e];
    y[i] = (s >= 0 && t >= 0) ? pos[3 * s + c] - pos[3 * t + c] : 0.f;
  }
}

// adjoint of the difference: out[i] = sum_{e: src_e = i} g[e] - sum_{e: dst_e = i} g[e]   (CSR by target + by-source view)
__global__ void dd_edge_scatter_kernel(const float* __restrict__ g, const int* __restrict__ rowptr,
                                       const int* __restrict__ rowptr_s, const int* __restrict__ perm_s, int N,
                                       float* __restrict__ y) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * N; i += gridDim.x * 256) {
    const int a = i / 3, c = i - 3 * a;
    float s = 0.f;
    for (int q = rowptr_s[a]; q < rowptr_s[a + 1]; ++q) s += g[3 * perm_s[q] + c];
    for (int e = rowptr[a]; e < rowptr[a + 1]; ++e) s -= g[3 * e + c];
    y[i] = s;
  }
}

// d[e] = |v[e]| (1 for padded slots: their distance never matters but must stay differentiable)
__global__ void dd_row_norm_kernel(const float* __restrict__ v, const int* __restrict__ src, int E, float* __restrict__ y) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < E; e += gridDim.x * 256) {
    const float a = v[3 * e], b = v[3 * e + 1], c = v[3 * e + 2];
    y[e] = src[e] >= 0 ? sqrtf((a * a + b * b) + c * c) : 1.f;
  }
}

// y[i] = g[batch[i]] * (mean ? 1 / max(count, 1) : 1)
__global__ void dd_seg_expand_kernel(const float* __restrict__ g, const int* __restrict__ batch, const int* __restrict__ mol_ptr,
                                     int N, int K, int mean, float* __restrict__ y) {
  const long long n = (long long)N * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int a = (int)(i / K), k = (int)(i - (long long)a * K), b = batch[a];
    const float sc = mean ? 1.f / (float)max(mol_ptr[b + 1] - mol_ptr[b], 1) : 1.f;
    y[i] = g[(size_t)b * K + k] * sc;
  }
}

__global__ void dd_broadcast_rows_kernel(const float* __restrict__ b, int M, int K, float* __restrict__ y) {
  const long long n = (long long)M * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = b[i % K];
}

// [E][3] <-> [3][E] (component-major copies of per-edge 3-vectors: every component is then a contiguous row vector)
__global__ void dd_transpose3_kernel(const float* __restrict__ x, int E, int to_soa, float* __restrict__ y) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * E; i += gridDim.x * 256) {
    const int e = i / 3, c = i - 3 * e;
    if (to_soa) y[(size_t)c * E + e] = x[i]; else y[i] = x[(size_t)c * E + e];
  }
}
__global__ void dd_merge3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, int E,
                                 float* __restrict__ y) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < E; e += gridDim.x * 256) {
    y[3 * e] = a ? a[e] : 0.f; y[3 * e + 1] = b ? b[e] : 0.f; y[3 * e + 2] = c ? c[e] : 0.f;
  }
}
extern "C" int msde_dd_merge3(const float* a, const float* b, const float* c, int E, float* y, void* stream) {
  if (E < 0 || !y) return MSDE_EINVAL;
  if (E == 0) return 0;
  MSDE_LAUNCH(dd_merge3_kernel, DD_GRID((long long)E), dim3(256), 0, as_stream(stream), a, b, c, E, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_dd_transpose3(const float* x, int E, int to_soa, float* y, void* stream) {
  if (E < 0 || !x || !y) return MSDE_EINVAL;
  if (E == 0) return 0;
  MSDE_LAUNCH(dd_transpose3_kernel, DD_GRID(3LL * E), dim3(256), 0, as_stream(stream), x, E, to_soa, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_dd_unary(const float* x, const int* mask, long long n, int kind, int order, float p0, float* y,
                             void* stream) {
  if (n < 0 || !x || !y || kind < 0 || kind > 4 || order < 0 || order > 2) return MSDE_EINVAL;
  if (n == 0) return 0;
  MSDE_LAUNCH(dd_unary_kernel, DD_GRID(n), dim3(256), 0, as_stream(stream), x, mask, n, kind, order, p0, (const float*)nullptr,
              (const float*)nullptr, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
// y = g * (g2 ? g2 : 1) * f^(order)(x): what the backward of msde_dd_unary (and of this operator) computes, as one launch
extern "C" int msde_dd_unary_mul(const float* g, const float* g2, const float* x, const int* mask, long long n, int kind,
                                 int order, float p0, float* y, void* stream) {
  if (n < 0 || !g || !x || !y || kind < 0 || kind > 4 || order < 0 || order > 2) return MSDE_EINVAL;
  if (n == 0) return 0;
  MSDE_LAUNCH(dd_unary_kernel, DD_GRID(n), dim3(256), 0, as_stream(stream), x, mask, n, kind, order, p0, g, g2, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_dd_rbf(const float* d, const int* src, const float* mu, int E, int G, float coeff, int order, float* y,
                           void* stream) {
  if (E < 0 || G <= 0 || !d || !mu || !y || order < 0 || order > 2) return MSDE_EINVAL;
  if (E == 0) return 0;
  MSDE_LAUNCH(dd_rbf_kernel, DD_GRID((long long)E * G), dim3(256), 0, as_stream(stream), d, src, mu, E, G, coeff, order, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_dd_binary(const float* a, const float* b, long long n, int op, float alpha, float* y, void* stream) {
  if (n < 0 || !a || !y || op < 0 || op > 2 || (op != 2 && !b)) return MSDE_EINVAL;
  if (n == 0) return 0;
  MSDE_LAUNCH(dd_binary_kernel, DD_GRID(n), dim3(256), 0, as_stream(stream), a, b, n, op, alpha, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_dd_sum_n(const float* const* srcs, int n, long long count, float* y, void* stream) {
  if (!srcs || n < 1 || n > 8 || count < 0 || !y) return MSDE_EINVAL;
  dd_ptrs8 s;
  for (int k = 0; k < 8; ++k) {
    s.p[k] = k < n ? srcs[k] : nullptr;
    if (k < n && !s.p[k]) return MSDE_EINVAL;
  }
  if (count == 0) return 0;
  MSDE_LAUNCH(dd_sum_n_kernel, DD_GRID(count), dim3(256), 0, as_stream(stream), s, n, count, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_dd_sum_rows_n(const float* const* srcs, const int* lds, int n, int rows, int cols, float* y, void* stream) {
  if (!srcs || !lds || n < 1 || n > 8 || rows < 0 || cols <= 0 || cols % 4 || !y) return MSDE_EINVAL;
  if ((reinterpret_cast<uintptr_t>(y) & 15) != 0) return MSDE_EINVAL;
  dd_ptrs8 s;
  dd_lds8 l;
  for (int k = 0; k < 8; ++k) {
    s.p[k] = k < n ? srcs[k] : nullptr;
    l.ld[k] = k < n ? lds[k] : 0;
    if (k < n && (!s.p[k] || lds[k] < cols || lds[k] % 4 || (reinterpret_cast<uintptr_t>(s.p[k]) & 15) != 0)) return MSDE_EINVAL;
  }
  if (rows == 0) return 0;
  MSDE_LAUNCH(dd_sum_rows_n_kernel, DD_GRID((long long)rows * (cols / 4)), dim3(256), 0, as_stream(stream), s, l, n, rows, cols / 4, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_dd_mul_rows(const float* M, const float* s, int E, int K, float* y, void* stream) {
  if (E < 0 || K <= 0 || !M || !s || !y) return MSDE_EINVAL;
  if (E == 0) return 0;
  MSDE_LAUNCH(dd_mul_rows_kernel, DD_GRID((long long)E * K), dim3(256), 0, as_stream(stream), M, s, E, K, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_dd_row_dot(const float* a, const float* b, int E, int K, float* y, void* stream) {
  if (E < 0 || K <= 0 || !a || !b || !y) return MSDE_EINVAL;
  if (E == 0) return 0;
  MSDE_LAUNCH(dd_row_dot_kernel, DD_GRID((long long)E * 64), dim3(256), 0, as_stream(stream), a, b, E, K, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_dd_edge_diff(const float* pos, const int* src, const int* dst, int E, float* y, void* stream) {
  if (E < 0 || !pos || !src || !dst || !y) return MSDE_EINVAL;
  if (E == 0) return 0;
  MSDE_LAUNCH(dd_edge_diff_kernel, DD_GRID(3LL * E), dim3(256), 0, as_stream(stream), pos, src, dst, E, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_dd_edge_scatter(const float* g, const int* rowptr, const int* rowptr_s, const int* perm_s, int N, float* y,
                                    void* stream) {
  if (N < 0 || !g || !rowptr || !rowptr_s || !perm_s || !y) return MSDE_EINVAL;
  if (N == 0) return 0;
  MSDE_LAUNCH(dd_edge_scatter_kernel, DD_GRID(3LL * N), dim3(256), 0, as_stream(stream), g, rowptr, rowptr_s, perm_s, N, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_dd_row_norm(const float* v, const int* src, int E, float* y, void* stream) {
  if (E < 0 || !v || !src || !y) return MSDE_EINVAL;
  if (E == 0) return 0;
  MSDE_LAUNCH(dd_row_norm_kernel, DD_GRID((long long)E), dim3(256), 0, as_stream(stream), v, src, E, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_dd_seg_expand(const float* g, const int* batch, const int* mol_ptr, int N, int K, int mean, float* y,
                                  void* stream) {
  if (N < 0 || K <= 0 || !g || !batch || !mol_ptr || !y) return MSDE_EINVAL;
  if (N == 0) return 0;
  MSDE_LAUNCH(dd_seg_expand_kernel, DD_GRID((long long)N * K), dim3(256), 0, as_stream(stream), g, batch, mol_ptr, N, K, mean, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_dd_broadcast_rows(const float* b, int M, int K, float* y, void* stream) {
  if (M < 0 || K <= 0 || !b || !y) return MSDE_EINVAL;
  if (M == 0) return 0;
  MSDE_LAUNCH(dd_broadcast_rows_kernel, DD_GRID((long long)M * K), dim3(256), 0, as_stream(stream), b, M, K, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
