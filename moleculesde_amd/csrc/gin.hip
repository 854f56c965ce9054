// gin.hip — GIN message passing with the bond-embedding sum fused in (molecule_gnn_model.py:13-32).
#include "msde_common.h"

template <int V>
__device__ __forceinline__ typename VecT<V>::type bond_emb(const typename VecT<V>::type* __restrict__ Tb,
                                                           const int* __restrict__ codes, int e, int cols, int c) {
  // left-to-right like BondEncoder.forward: ((0 + t0) + t1) + t2
  auto a = Tb[(size_t)codes[3 * e] * cols + c];
  a = vadd(a, Tb[(size_t)codes[3 * e + 1] * cols + c]);
  a = vadd(a, Tb[(size_t)codes[3 * e + 2] * cols + c]);
  return a;
}

template <int V>
__global__ void gin_aggregate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ tab,
                                         const int* __restrict__ codes, const float* __restrict__ eps,
                                         const int* __restrict__ rowptr, const int* __restrict__ src, int N, int cols,
                                         int tpr, float* __restrict__ out) {
  using T = typename VecT<V>::type;
  int rpb = blockDim.x / tpr;
  int i = blockIdx.x * rpb + threadIdx.x / tpr;
  int lane = threadIdx.x % tpr;
  if (i >= N) return;
  const T* X = reinterpret_cast<const T*>(x);
  const T* Tb = reinterpret_cast<const T*>(tab);
  T* O = reinterpret_cast<T*>(out);
  float ope = 1.f + eps[0];
  int s0 = rowptr[i], s1 = rowptr[i + 1];
  for (int c = lane; c < cols; c += tpr) {
    T acc = vzero<V>();
    for (int e = s0; e < s1; ++e) {
      T m = vadd(X[(size_t)src[e] * cols + c], bond_emb<V>(Tb, codes, e, cols, c));
      acc = vadd(acc, vrelu(m));
    }
    O[(size_t)i * cols + c] = vadd(vscale(X[(size_t)i * cols + c], ope), acc);
  }
}

template <int V>
__global__ void gin_aggregate_bwd_x_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                           const float* __restrict__ tab, const int* __restrict__ codes,
                                           const float* __restrict__ eps, const int* __restrict__ rowptr_s,
                                           const int* __restrict__ perm_s, const int* __restrict__ dst, int N,
                                           int cols, int tpr, float* __restrict__ g_x) {
  using T = typename VecT<V>::type;
  int rpb = blockDim.x / tpr;
  int j = blockIdx.x * rpb + threadIdx.x / tpr;
  int lane = threadIdx.x % tpr;
  if (j >= N) return;
  const T* X = reinterpret_cast<const T*>(x);
  const T* G = reinterpret_cast<const T*>(g);
  const T* Tb = reinterpret_cast<const T*>(tab);
  T* O = reinterpret_cast<T*>(g_x);
  float ope = 1.f + eps[0];
  int s0 = rowptr_s[j], s1 = rowptr_s[j + 1];
  for (int c = lane; c < cols; c += tpr) {
    T xj = X[(size_t)j * cols + c];
    T acc = vscale(G[(size_t)j * cols + c], ope);
    for (int s = s0; s < s1; ++s) {
      int e = perm_s[s];
      T m = vadd(xj, bond_emb<V>(Tb, codes, e, cols, c));
      acc = vadd(acc, vgate(G[(size_t)dst[e] * cols + c], m));
    }
    O[(size_t)j * cols + c] = acc;
  }
}

// by-target pass: table gradient accumulated in LDS (R*D floats), flushed with global atomics;
// g_eps = sum_i g[i].x[i] reduced per block then one atomic.
__global__ void gin_aggregate_bwd_tab_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                             const float* __restrict__ tab, const int* __restrict__ codes,
                                             const int* __restrict__ rowptr, const int* __restrict__ src, int N, int D,
                                             int R, int nodes_per_block, float* __restrict__ g_tab,
                                             float* __restrict__ g_eps) {
  extern __shared__ float lds[];  // [R*D] + [blockDim/64]
  float* ltab = lds;
  float* red = lds + (size_t)R * D;
  for (int t = threadIdx.x; t < R * D; t += blockDim.x) ltab[t] = 0.f;
  __syncthreads();
  int n0 = blockIdx.x * nodes_per_block, n1 = min(n0 + nodes_per_block, N);
  float eacc = 0.f;
  for (int i = n0; i < n1; ++i) {
    int s0 = rowptr[i], s1 = rowptr[i + 1];
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
      float gi = g[(size_t)i * D + c];
      eacc = fmaf(gi, x[(size_t)i * D + c], eacc);
      for (int e = s0; e < s1; ++e) {
        int c0 = codes[3 * e], c1 = codes[3 * e + 1], c2 = codes[3 * e + 2];
        float emb = (tab[(size_t)c0 * D + c] + tab[(size_t)c1 * D + c]) + tab[(size_t)c2 * D + c];
        float m = x[(size_t)src[e] * D + c] + emb;
        if (m > 0.f) {
          // column c is owned by this thread within the block: no LDS race, plain read-modify-write
          ltab[c0 * D + c] += gi;
          ltab[c1 * D + c] += gi;
          ltab[c2 * D + c] += gi;
        }
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < R * D; t += blockDim.x) {
    float v = ltab[t];
    if (v != 0.f) atomicAdd(&g_tab[t], v);
  }
  // block reduce eacc
  eacc = group_sum(eacc, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = eacc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    atomicAdd(g_eps, s);
  }
}

extern "C" int msde_gin_aggregate_fwd(const float* x, const float* tab, const int* codes, const float* eps,
                                      const int* rowptr, const int* src, int N, int D, float* out, void* stream) {
  if (N < 0 || D <= 0 || !x || !tab || !eps || !rowptr || !out) return MSDE_EINVAL;
  if (N == 0) return 0;
  LAUNCH_ROWS(gin_aggregate_fwd_kernel, N, D, x, tab, codes, eps, rowptr, src, N, cols, tpr, out);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_gin_aggregate_bwd_x(const float* g, const float* x, const float* tab, const int* codes,
                                        const float* eps, const int* rowptr_s, const int* perm_s, const int* dst,
                                        int N, int D, float* g_x, void* stream) {
  if (N < 0 || D <= 0 || !g || !x || !tab || !eps || !rowptr_s || !g_x) return MSDE_EINVAL;
  if (N == 0) return 0;
  LAUNCH_ROWS(gin_aggregate_bwd_x_kernel, N, D, g, x, tab, codes, eps, rowptr_s, perm_s, dst, N, cols, tpr, g_x);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_gin_aggregate_bwd_tab(const float* g, const float* x, const float* tab, const int* codes,
                                          const int* rowptr, const int* src, int N, int D, int R, float* g_tab,
                                          float* g_eps, void* stream) {
  if (N < 0 || D <= 0 || R <= 0 || !g || !x || !tab || !rowptr || !g_tab || !g_eps) return MSDE_EINVAL;
  size_t lds = ((size_t)R * D + 8) * sizeof(float);
  if (lds > 64 * 1024) return MSDE_EUNSUP;
  if (N == 0) return 0;
  int npb = 16;
  MSDE_LAUNCH(gin_aggregate_bwd_tab_kernel, dim3((N + npb - 1) / npb), dim3(256), lds, as_stream(stream), g, x,
                     tab, codes, rowptr, src, N, D, R, npb, g_tab, g_eps);
  MSDE_CHECK_LAUNCH();
  return 0;
}
