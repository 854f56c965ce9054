// graph.hip — radius-graph CSR construction, scans, generic CSR row ops, embedding sums.
// gfx950 only.  See include/msde_hip.h for the contract of each entry point.
#include "msde_common.h"

// ------------------------------------------------------------------------------------------------
// radius graph (torch_cluster.radius semantics): per target i scan the atoms j of
// its own molecule in index order; strict |pi-pj|^2 < r2; stop after max_nbr hits.
// One thread per target atom: molecules are <= a few dozen atoms, the scan is a handful of
// L1/L2-resident loads; the work is O(sum n_m^2) distance tests.
// ------------------------------------------------------------------------------------------------
__global__ void radius_count_kernel(const float* __restrict__ pos, const int* __restrict__ batch,
                                    const int* __restrict__ mol_ptr, int N, float r2, int max_nbr,
                                    int* __restrict__ deg) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int m = batch[i];
  int j0 = mol_ptr[m], j1 = mol_ptr[m + 1];
  float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
  int cnt = 0;
  for (int j = j0; j < j1 && cnt < max_nbr; ++j) {
    if (j == i) continue;
    float dx = xi - pos[3 * j], dy = yi - pos[3 * j + 1], dz = zi - pos[3 * j + 2];
    // same association as torch: ((dx*dx + dy*dy) + dz*dz), no fma contraction surprises matter
    float d2 = dx * dx + dy * dy + dz * dz;
    cnt += (d2 < r2) ? 1 : 0;
  }
  deg[i] = cnt;
}

__global__ void radius_fill_kernel(const float* __restrict__ pos, const int* __restrict__ batch,
                                   const int* __restrict__ mol_ptr, int N, float r2, int max_nbr,
                                   const int* __restrict__ rowptr, int* __restrict__ src,
                                   int* __restrict__ dst, float* __restrict__ dist, int E_cap) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    int m = batch[i];
    int j0 = mol_ptr[m], j1 = mol_ptr[m + 1];
    float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    int slot = rowptr[i];
    int cnt = 0;
    for (int j = j0; j < j1 && cnt < max_nbr; ++j) {
      if (j == i) continue;
      float dx = xi - pos[3 * j], dy = yi - pos[3 * j + 1], dz = zi - pos[3 * j + 2];
      float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < r2) {
        if (slot < E_cap) {
          src[slot] = j;
          dst[slot] = i;
          dist[slot] = sqrtf(d2);
        }
        ++slot;
        ++cnt;
      }
    }
  }
  // padded tail: grid-stride over [rowptr[N], E_cap)
  int E = rowptr[N];
  for (int e = E + blockIdx.x * blockDim.x + threadIdx.x; e < E_cap; e += gridDim.x * blockDim.x) {
    src[e] = -1;
    dst[e] = -1;
    dist[e] = 0.f;
  }
}

// single-workgroup exclusive scan (n up to ~16M, 1024 threads, chunked with a running carry)
__global__ void __launch_bounds__(1024) exclusive_scan_kernel(const int* __restrict__ in, int* __restrict__ out, int n) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int idx = base + tid;
    int v = idx < n ? in[idx] : 0;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; ++w) woff += wsum[w];
    int carry = carry_s;
    if (idx < n) out[idx] = carry + woff + incl - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (tid == 0) out[n] = carry_s;
}

extern "C" int msde_radius_count(const float* pos, const int* batch, const int* mol_ptr, int N, float r2,
                                 int max_nbr, int* deg, void* stream) {
  if (N < 0 || (N > 0 && (!pos || !batch || !mol_ptr || !deg))) return MSDE_EINVAL;
  if (N == 0) return 0;
  MSDE_LAUNCH(radius_count_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), pos, batch,
                     mol_ptr, N, r2, max_nbr, deg);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_exclusive_scan_i32(const int* in, int* out, int n, void* stream) {
  if (n < 0 || !out || (n > 0 && !in)) return MSDE_EINVAL;
  MSDE_LAUNCH(exclusive_scan_kernel, dim3(1), dim3(1024), 0, as_stream(stream), in, out, n);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_radius_fill(const float* pos, const int* batch, const int* mol_ptr, int N, float r2,
                                int max_nbr, const int* rowptr, int* src, int* dst, float* dist, int E_cap,
                                void* stream) {
  if (N < 0 || E_cap < 0 || !rowptr) return MSDE_EINVAL;
  if (N == 0 && E_cap == 0) return 0;
  int work = N > 0 ? N : 1;
  MSDE_LAUNCH(radius_fill_kernel, dim3((work + 255) / 256), dim3(256), 0, as_stream(stream), pos, batch,
                     mol_ptr, N, r2, max_nbr, rowptr, src, dst, dist, E_cap);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// generic CSR row ops.  Thread layout for every "row" kernel in this library: a group of TPR
// consecutive lanes (power of two <= 64) owns one row and walks its float4 columns with stride TPR,
// so a wave reads 1 KiB-contiguous pieces of each gathered row.
// ------------------------------------------------------------------------------------------------
template <int V>
__global__ void segment_sum_rows_kernel(const float* __restrict__ rows, const int* __restrict__ rowptr,
                                        const int* __restrict__ perm, int N, int cols, int tpr, float mean,
                                        float* __restrict__ out, int ldo_cols) {
  using T = typename VecT<V>::type;
  int rpb = blockDim.x / tpr;
  int i = blockIdx.x * rpb + threadIdx.x / tpr;
  int lane = threadIdx.x % tpr;
  if (i >= N) return;
  int s0 = rowptr[i], s1 = rowptr[i + 1];
  float scale = 1.f;
  if (mean != 0.f) scale = 1.f / (float)max(s1 - s0, 1);
  const T* R = reinterpret_cast<const T*>(rows);
  T* O = reinterpret_cast<T*>(out);
  for (int c = lane; c < cols; c += tpr) {
    T acc = vzero<V>();
    for (int s = s0; s < s1; ++s) {
      int e = perm ? perm[s] : s;
      acc = vadd(acc, R[(size_t)e * cols + c]);
    }
    O[(size_t)i * ldo_cols + c] = vscale(acc, scale);
  }
}

extern "C" int msde_segment_sum_rows(const float* rows, const int* rowptr, const int* perm, int N, int D,
                                     float scale_by_inv_count, float* out, int ldo, void* stream) {
  if (N < 0 || D <= 0 || !rowptr || !out) return MSDE_EINVAL;
  if (ldo <= 0) ldo = D;
  if (ldo < D) return MSDE_EINVAL;
  if (N == 0) return 0;
  if (D % 4 == 0 && ldo % 4 == 0) {
    int cols = D / 4, tpr = pick_tpr(cols), rpb = 256 / tpr;
    MSDE_LAUNCH(segment_sum_rows_kernel<4>, dim3((N + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), rows,
                       rowptr, perm, N, cols, tpr, scale_by_inv_count, out, ldo / 4);
  } else {
    int cols = D, tpr = pick_tpr(cols), rpb = 256 / tpr;
    MSDE_LAUNCH(segment_sum_rows_kernel<1>, dim3((N + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), rows,
                       rowptr, perm, N, cols, tpr, scale_by_inv_count, out, ldo);
  }
  MSDE_CHECK_LAUNCH();
  return 0;
}

template <int V>
__global__ void pair_gather_add_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                       const int* __restrict__ src, const int* __restrict__ dst, int E, int cols,
                                       int tpr, float* __restrict__ out) {
  using T = typename VecT<V>::type;
  int rpb = blockDim.x / tpr;
  int e = blockIdx.x * rpb + threadIdx.x / tpr;
  int lane = threadIdx.x % tpr;
  if (e >= E) return;
  int j = src[e], i = dst ? dst[e] : -1;
  T* O = reinterpret_cast<T*>(out) + (size_t)e * cols;
  if (j < 0) {
    for (int c = lane; c < cols; c += tpr) O[c] = vzero<V>();
    return;
  }
  const T* a = reinterpret_cast<const T*>(A) + (size_t)j * cols;
  if (B) {
    const T* b = reinterpret_cast<const T*>(B) + (size_t)i * cols;
    for (int c = lane; c < cols; c += tpr) O[c] = vadd(a[c], b[c]);
  } else {
    for (int c = lane; c < cols; c += tpr) O[c] = a[c];
  }
}

extern "C" int msde_pair_gather_add(const float* A, const float* B, const int* src, const int* dst, int E, int D,
                                    float* out, void* stream) {
  if (E < 0 || D <= 0 || !A || !B || !src || !dst || !out) return MSDE_EINVAL;
  if (E == 0) return 0;
  if (D % 4 == 0) {
    int cols = D / 4, tpr = pick_tpr(cols), rpb = 256 / tpr;
    MSDE_LAUNCH(pair_gather_add_kernel<4>, dim3((E + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), A, B,
                       src, dst, E, cols, tpr, out);
  } else {
    int cols = D, tpr = pick_tpr(cols), rpb = 256 / tpr;
    MSDE_LAUNCH(pair_gather_add_kernel<1>, dim3((E + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), A, B,
                       src, dst, E, cols, tpr, out);
  }
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_gather_rows(const float* X, const int* idx, int E, int D, float* out, void* stream) {
  if (E < 0 || D <= 0 || !X || !idx || !out) return MSDE_EINVAL;
  if (E == 0) return 0;
  if (D % 4 == 0) {
    int cols = D / 4, tpr = pick_tpr(cols), rpb = 256 / tpr;
    MSDE_LAUNCH(pair_gather_add_kernel<4>, dim3((E + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), X,
                       (const float*)nullptr, idx, (const int*)nullptr, E, cols, tpr, out);
  } else {
    int cols = D, tpr = pick_tpr(cols), rpb = 256 / tpr;
    MSDE_LAUNCH(pair_gather_add_kernel<1>, dim3((E + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), X,
                       (const float*)nullptr, idx, (const int*)nullptr, E, cols, tpr, out);
  }
  MSDE_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// embedding sums (ogb AtomEncoder / BondEncoder / nn.Embedding)
// ------------------------------------------------------------------------------------------------
template <int V>
__global__ void embedding_sum_fwd_kernel(const float* __restrict__ tab, const int* __restrict__ codes, int N, int K,
                                         int cols, int tpr, float* __restrict__ out) {
  using T = typename VecT<V>::type;
  int rpb = blockDim.x / tpr;
  int i = blockIdx.x * rpb + threadIdx.x / tpr;
  int lane = threadIdx.x % tpr;
  if (i >= N) return;
  const T* Tb = reinterpret_cast<const T*>(tab);
  T* O = reinterpret_cast<T*>(out) + (size_t)i * cols;
  for (int c = lane; c < cols; c += tpr) {
    // same left-to-right order as the reference's  x_embedding += emb_k(x[:,k])  (starts from 0)
    T acc = Tb[(size_t)codes[(size_t)i * K] * cols + c];
    for (int k = 1; k < K; ++k) acc = vadd(acc, Tb[(size_t)codes[(size_t)i * K + k] * cols + c]);
    O[c] = acc;
  }
}

extern "C" int msde_embedding_sum_fwd(const float* tab, const int* codes, int N, int K, int D, float* out,
                                      void* stream) {
  if (N < 0 || K <= 0 || D <= 0 || !tab || !codes || !out) return MSDE_EINVAL;
  if (N == 0) return 0;
  if (D % 4 == 0) {
    int cols = D / 4, tpr = pick_tpr(cols), rpb = 256 / tpr;
    MSDE_LAUNCH(embedding_sum_fwd_kernel<4>, dim3((N + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), tab,
                       codes, N, K, cols, tpr, out);
  } else {
    int cols = D, tpr = pick_tpr(cols), rpb = 256 / tpr;
    MSDE_LAUNCH(embedding_sum_fwd_kernel<1>, dim3((N + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), tab,
                       codes, N, K, cols, tpr, out);
  }
  MSDE_CHECK_LAUNCH();
  return 0;
}

// block (r, s): table row r, slice s of its node list; one wave-wide pass over D per slice.
__global__ void embedding_sum_bwd_kernel(const float* __restrict__ g, const int* __restrict__ list_ptr,
                                         const int* __restrict__ list_nodes, int D, int split,
                                         float* __restrict__ g_tab) {
  int r = blockIdx.x, s = blockIdx.y;
  int p0 = list_ptr[r], p1 = list_ptr[r + 1];
  int len = p1 - p0;
  if (len == 0) return;
  // lists shorter than 32 rows stay in one slice (plain store, bitwise deterministic)
  int chunk = max((len + split - 1) / split, 32);
  int a = p0 + s * chunk, b = min(a + chunk, p1);
  if (a >= b) return;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float acc = 0.f;
    for (int p = a; p < b; ++p) acc += g[(size_t)list_nodes[p] * D + c];
    if (len <= chunk)
      g_tab[(size_t)r * D + c] = acc;  // single slice owns the row: plain store, deterministic
    else
      atomicAdd(&g_tab[(size_t)r * D + c], acc);
  }
}

extern "C" int msde_embedding_sum_bwd(const float* g, const int* list_ptr, const int* list_nodes, int R, int D,
                                      int split, float* g_tab, void* stream) {
  if (R <= 0 || D <= 0 || split <= 0 || !g || !list_ptr || !list_nodes || !g_tab) return MSDE_EINVAL;
  int threads = D >= 256 ? 256 : ((D + 63) / 64) * 64;
  MSDE_LAUNCH(embedding_sum_bwd_kernel, dim3(R, split), dim3(threads), 0, as_stream(stream), g, list_ptr,
                     list_nodes, D, split, g_tab);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_abi_version(void) { return 1; }
extern "C" const char* msde_target_arch(void) { return "gfx950"; }
