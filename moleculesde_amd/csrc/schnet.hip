// schnet.hip — SchNet edge kernels, v1 (decomposed): Gaussian smearing + cosine cutoff, and the
// CFConv gather * filter -> segmented sum with its two backward passes (schnet.py:185-207).
#include "msde_common.h"

__global__ void rbf_cutoff_fwd_kernel(const float* __restrict__ dist, const int* __restrict__ E_dev, int E_cap, int G,
                                      const float* __restrict__ offset, float coeff, float cutoff,
                                      float* __restrict__ rbf, float* __restrict__ C) {
  const float PI_F = 3.14159265358979323846f;
  int E = E_dev ? E_dev[0] : E_cap;
  size_t total = (size_t)E_cap * G;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    int e = (int)(t / G), gidx = (int)(t % G);
    float v = 0.f;
    if (e < E) {
      float d = dist[e];
      float diff = d - offset[gidx];
      v = expf(coeff * (diff * diff));
    }
    rbf[t] = v;
  }
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E_cap; e += gridDim.x * blockDim.x) {
    float c = 0.f;
    if (e < E) c = 0.5f * (cosf(dist[e] * PI_F / cutoff) + 1.0f);
    C[e] = c;
  }
}

extern "C" int msde_rbf_cutoff_fwd(const float* dist, const int* E_dev, int E_cap, int G, const float* offset,
                                   float coeff, float cutoff, float* rbf, float* C, void* stream) {
  if (E_cap < 0 || G <= 0 || !dist || !offset || !rbf || !C) return MSDE_EINVAL;
  if (E_cap == 0) return 0;
  size_t total = (size_t)E_cap * G;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  MSDE_LAUNCH(rbf_cutoff_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dist, E_dev, E_cap, G,
                     offset, coeff, cutoff, rbf, C);
  MSDE_CHECK_LAUNCH();
  return 0;
}

template <int V>
__global__ void cfconv_aggregate_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ Wf,
                                            const float* __restrict__ C, const int* __restrict__ rowptr,
                                            const int* __restrict__ src, int N, int cols, int tpr,
                                            float* __restrict__ agg) {
  using T = typename VecT<V>::type;
  int rpb = blockDim.x / tpr;
  int i = blockIdx.x * rpb + threadIdx.x / tpr;
  int lane = threadIdx.x % tpr;
  if (i >= N) return;
  const T* X = reinterpret_cast<const T*>(x1);
  const T* W = reinterpret_cast<const T*>(Wf);
  T* O = reinterpret_cast<T*>(agg);
  int s0 = rowptr[i], s1 = rowptr[i + 1];
  for (int c = lane; c < cols; c += tpr) {
    T acc = vzero<V>();
    int e = s0;
    // batches of 8 edges: the indices, then all 16 rows in flight, accumulated in edge order (one edge at a time this loop is a
    // chain of index -> row round trips: 12 us for the 20 in-edges of an MD17 atom)
    for (; e + 7 < s1; e += 8) {
      int j[8];
      T w[8], x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) j[u] = src[e + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) { w[u] = W[(size_t)(e + u) * cols + c]; x[u] = X[(size_t)j[u] * cols + c]; }
      if (C) {
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = vscale(w[u], C[e + u]);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = vadd(acc, vmul(x[u], w[u]));
    }
    for (; e + 3 < s1; e += 4) {
      int j[4];
      T w[4], x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) j[u] = src[e + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) { w[u] = W[(size_t)(e + u) * cols + c]; x[u] = X[(size_t)j[u] * cols + c]; }
      if (C) {
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = vscale(w[u], C[e + u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = vadd(acc, vmul(x[u], w[u]));
    }
    for (; e < s1; ++e) {
      // message = x_j * (nn(edge_attr) * C): keep the reference's rounding order (W*C first)
      T w = C ? vscale(W[(size_t)e * cols + c], C[e]) : W[(size_t)e * cols + c];
      acc = vadd(acc, vmul(X[(size_t)src[e] * cols + c], w));
    }
    O[(size_t)i * cols + c] = acc;
  }
}

template <int V>
__global__ void cfconv_aggregate_bwd_w_kernel(const float* __restrict__ g_agg, const float* __restrict__ x1,
                                              const float* __restrict__ C, const int* __restrict__ rowptr,
                                              const int* __restrict__ src, int N, int cols, int tpr, int E_cap,
                                              float* __restrict__ g_Wf) {
  using T = typename VecT<V>::type;
  int rpb = blockDim.x / tpr;
  int i = blockIdx.x * rpb + threadIdx.x / tpr;
  int lane = threadIdx.x % tpr;
  const T* X = reinterpret_cast<const T*>(x1);
  const T* G = reinterpret_cast<const T*>(g_agg);
  T* O = reinterpret_cast<T*>(g_Wf);
  if (i < N) {
    int s0 = rowptr[i], s1 = rowptr[i + 1];
    for (int c = lane; c < cols; c += tpr) {
      T gi = G[(size_t)i * cols + c];
      int e = s0;
      for (; e + 7 < s1; e += 8) {             // batches of 8 edges: indices, then the 8 rows in flight
        int j[8];
        T x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) j[u] = src[e + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = X[(size_t)j[u] * cols + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          T v = vmul(gi, x[u]);
          O[(size_t)(e + u) * cols + c] = C ? vscale(v, C[e + u]) : v;
        }
      }
      for (; e < s1; ++e) {
        T v = vmul(gi, X[(size_t)src[e] * cols + c]);
        O[(size_t)e * cols + c] = C ? vscale(v, C[e]) : v;
      }
    }
  }
  // zero the padded tail rows [rowptr[N], E_cap): they feed the weight-gradient GEMMs
  int E = rowptr[N];
  int row0 = E + blockIdx.x * rpb + threadIdx.x / tpr;
  for (int e = row0; e < E_cap; e += gridDim.x * rpb)
    for (int c = lane; c < cols; c += tpr) O[(size_t)e * cols + c] = vzero<V>();
}

template <int V>
__global__ void cfconv_aggregate_bwd_x_kernel(const float* __restrict__ g_agg, const float* __restrict__ Wf,
                                              const float* __restrict__ C, const int* __restrict__ rowptr_s,
                                              const int* __restrict__ perm_s, const int* __restrict__ dst, int N,
                                              int cols, int tpr, float* __restrict__ g_x1) {
  using T = typename VecT<V>::type;
  int rpb = blockDim.x / tpr;
  int j = blockIdx.x * rpb + threadIdx.x / tpr;
  int lane = threadIdx.x % tpr;
  if (j >= N) return;
  const T* W = reinterpret_cast<const T*>(Wf);
  const T* G = reinterpret_cast<const T*>(g_agg);
  T* O = reinterpret_cast<T*>(g_x1);
  int s0 = rowptr_s[j], s1 = rowptr_s[j + 1];
  for (int c = lane; c < cols; c += tpr) {
    T acc = vzero<V>();
    int s = s0;
    for (; s + 3 < s1; s += 4) {      // indices first, then the eight row loads in flight; accumulated in edge order
      int e0 = perm_s[s], e1 = perm_s[s + 1], e2 = perm_s[s + 2], e3 = perm_s[s + 3];
      int d0 = dst[e0], d1 = dst[e1], d2 = dst[e2], d3 = dst[e3];
      T w0 = W[(size_t)e0 * cols + c], w1 = W[(size_t)e1 * cols + c], w2 = W[(size_t)e2 * cols + c], w3 = W[(size_t)e3 * cols + c];
      T g0 = G[(size_t)d0 * cols + c], g1 = G[(size_t)d1 * cols + c], g2 = G[(size_t)d2 * cols + c], g3 = G[(size_t)d3 * cols + c];
      if (C) { w0 = vscale(w0, C[e0]); w1 = vscale(w1, C[e1]); w2 = vscale(w2, C[e2]); w3 = vscale(w3, C[e3]); }
      acc = vadd(acc, vmul(g0, w0)); acc = vadd(acc, vmul(g1, w1));
      acc = vadd(acc, vmul(g2, w2)); acc = vadd(acc, vmul(g3, w3));
    }
    for (; s < s1; ++s) {
      int e = perm_s[s];
      T w = C ? vscale(W[(size_t)e * cols + c], C[e]) : W[(size_t)e * cols + c];
      acc = vadd(acc, vmul(G[(size_t)dst[e] * cols + c], w));
    }
    O[(size_t)j * cols + c] = acc;
  }
}

extern "C" int msde_cfconv_aggregate_fwd(const float* x1, const float* Wf, const float* C, const int* rowptr,
                                         const int* src, int N, int F, float* agg, void* stream) {
  if (N < 0 || F <= 0 || !x1 || !Wf || !rowptr || !src || !agg) return MSDE_EINVAL;
  if (N == 0) return 0;
  LAUNCH_ROWS(cfconv_aggregate_fwd_kernel, N, F, x1, Wf, C, rowptr, src, N, cols, tpr, agg);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_cfconv_aggregate_bwd_w(const float* g_agg, const float* x1, const float* C, const int* rowptr,
                                           const int* src, int N, int F, int E_cap, float* g_Wf, void* stream) {
  if (N < 0 || F <= 0 || !g_agg || !x1 || !rowptr || !src || !g_Wf) return MSDE_EINVAL;
  if (N == 0) return 0;
  LAUNCH_ROWS(cfconv_aggregate_bwd_w_kernel, N, F, g_agg, x1, C, rowptr, src, N, cols, tpr, E_cap, g_Wf);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_cfconv_aggregate_bwd_x(const float* g_agg, const float* Wf, const float* C, const int* rowptr_s,
                                           const int* perm_s, const int* dst, int N, int F, float* g_x1,
                                           void* stream) {
  if (N < 0 || F <= 0 || !g_agg || !Wf || !rowptr_s || !perm_s || !dst || !g_x1) return MSDE_EINVAL;
  if (N == 0) return 0;
  LAUNCH_ROWS(cfconv_aggregate_bwd_x_kernel, N, F, g_agg, Wf, C, rowptr_s, perm_s, dst, N, cols, tpr, g_x1);
  MSDE_CHECK_LAUNCH();
  return 0;
}
