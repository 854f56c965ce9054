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

// ---- transposed (by-source) view of the radius graph, without a sort ---------------------------------------
// Edges never leave a molecule and every target row lists its sources in ascending order (radius_fill), so the
// edges of source j, in canonical order, are found by walking the molecule's target rows and binary-searching j
// in each: a stable counting sort with one thread per source, no atomics.  Pass 0 counts, pass 1 writes perm_s.
template <bool FILL>
__global__ void radius_transpose_kernel(const int* __restrict__ batch, const int* __restrict__ mol_ptr,
                                        const int* __restrict__ rowptr, const int* __restrict__ src, int N, int E_cap,
                                        int* __restrict__ deg_s, const int* __restrict__ rowptr_s,
                                        int* __restrict__ perm_s) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (FILL) {   // padded slots keep their own index (never read through rowptr_s)
    int E = rowptr[N];
    for (int e = E + j; e < E_cap; e += gridDim.x * blockDim.x) perm_s[e] = e;
  }
  if (j >= N) return;
  int m = batch[j];
  int i0 = mol_ptr[m], i1 = mol_ptr[m + 1];
  int cnt = 0;
  int base = FILL ? rowptr_s[j] : 0;
  for (int i = i0; i < i1; ++i) {
    int lo = rowptr[i], hi = rowptr[i + 1];
    while (lo < hi) {                       // first position with src >= j
      int mid = (lo + hi) >> 1;
      if (src[mid] < j) lo = mid + 1; else hi = mid;
    }
    if (lo < rowptr[i + 1] && src[lo] == j) {
      if (FILL) perm_s[base + cnt] = lo;
      ++cnt;
    }
  }
  if (!FILL) deg_s[j] = cnt;
}

extern "C" int msde_radius_transpose(const int* batch, const int* mol_ptr, const int* rowptr, const int* src, int N,
                                     int E_cap, int* deg_s, int* rowptr_s, int* perm_s, void* stream) {
  if (N < 0 || E_cap < 0 || !rowptr || !rowptr_s || (N > 0 && (!batch || !mol_ptr || !deg_s))) return MSDE_EINVAL;
  if (E_cap > 0 && (!src || !perm_s)) return MSDE_EINVAL;
  hipStream_t st = as_stream(stream);
  int work = N > 0 ? N : 1;
  dim3 grid((work + 255) / 256);
  MSDE_LAUNCH(radius_transpose_kernel<false>, grid, dim3(256), 0, st, batch, mol_ptr, rowptr, src, N, E_cap, deg_s,
              (const int*)nullptr, (int*)nullptr);
  MSDE_CHECK_LAUNCH();
  int rc = msde_exclusive_scan_i32(deg_s, rowptr_s, N, stream);
  if (rc != 0) return rc;
  MSDE_LAUNCH(radius_transpose_kernel<true>, grid, dim3(256), 0, st, batch, mol_ptr, rowptr, src, N, E_cap, deg_s,
              (const int*)rowptr_s, perm_s);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// The same view in ONE launch, one workgroup per molecule (<= RT_NMAX atoms): the edges of a molecule occupy the SAME contiguous
// range in both views (all edges are intra-molecular and atoms are grouped per molecule), so rowptr_s of its atoms starts at
// rowptr[first atom] and needs no scan over the batch.  Step 1: every (target i, source j) pair of the molecule looks j up in
// row i (binary search) -> its edge position or -1, in LDS; step 2: source j walks its column (targets ascending: the canonical
// order), counts, the counts are scanned inside the workgroup, and the same walk writes perm_s.  The three launches of
// msde_radius_transpose cost 16 + 4 + 16 us on the 21-atom MD17 graph (one thread per source, ~100 dependent loads each).
#define RT_NMAX 64
__global__ void __launch_bounds__(256)
radius_transpose_mol_kernel(const int* __restrict__ mol_ptr, int B, const int* __restrict__ rowptr, const int* __restrict__ src,
                            int N, int E_cap, int* __restrict__ rowptr_s, int* __restrict__ perm_s) {
  __shared__ int pos[RT_NMAX * RT_NMAX];
  __shared__ int cnt[RT_NMAX + 1];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int i0 = mol_ptr[m], i1 = mol_ptr[m + 1], n = i1 - i0;
  if (m == 0) {      // padded slots keep their own index (never read through rowptr_s); the closing row pointer, and the
    const int E = rowptr[N];                 // rows of atoms behind the last molecule (a capacity bucket's padding): empty
    for (int e = E + tid; e < E_cap; e += 256) perm_s[e] = e;
    for (int j = mol_ptr[B] + tid; j <= N; j += 256) rowptr_s[j] = E;
  }
  if (n <= 0 || n > RT_NMAX) return;          // (n > RT_NMAX: the host does not route such batches here)
  for (int idx = tid; idx < n * n; idx += 256) {
    const int i = idx / n, j = i0 + idx % n;
    int lo = rowptr[i0 + i];
    const int end = rowptr[i0 + i + 1];
    int hi = end;
    while (lo < hi) {                       // first position with src >= j
      const int mid = (lo + hi) >> 1;
      if (src[mid] < j) lo = mid + 1; else hi = mid;
    }
    pos[idx] = (lo < end && src[lo] == j) ? lo : -1;
  }
  __syncthreads();
  if (tid < n) {
    int c = 0;
    for (int i = 0; i < n; ++i) c += pos[i * n + tid] >= 0;
    cnt[tid] = c;
  }
  __syncthreads();
  if (tid == 0) {                             // exclusive scan of <= 64 counts
    int run = rowptr[i0];
    for (int j = 0; j < n; ++j) { const int c = cnt[j]; cnt[j] = run; run += c; }
  }
  __syncthreads();
  if (tid < n) {
    int base = cnt[tid];
    rowptr_s[i0 + tid] = base;
    for (int i = 0; i < n; ++i) {
      const int p = pos[i * n + tid];
      if (p >= 0) perm_s[base++] = p;
    }
  }
}

extern "C" int msde_radius_transpose_mol(const int* mol_ptr, int B, int n_max, const int* rowptr, const int* src, int N,
                                         int E_cap, int* rowptr_s, int* perm_s, void* stream) {
  if (B < 0 || N < 0 || E_cap < 0 || !mol_ptr || !rowptr || !rowptr_s) return MSDE_EINVAL;
  if (E_cap > 0 && (!src || !perm_s)) return MSDE_EINVAL;
  if (n_max > RT_NMAX || B == 0) return MSDE_EUNSUP;
  MSDE_LAUNCH(radius_transpose_mol_kernel, dim3(B), dim3(256), 0, as_stream(stream), mol_ptr, B, rowptr, src, N, E_cap, rowptr_s,
              perm_s);
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
                                        const int* __restrict__ perm, const int* __restrict__ rowptr2,
                                        const int* __restrict__ perm2, int N, int cols, int ldi_cols, int tpr, int epl,
                                        float mean, float* __restrict__ out, int ldo_cols) {
  // a group of tpr * epl lanes owns one output row: tpr column lanes x epl EDGE lanes (edge lane l sums slots l,
  // l + epl, ...; the epl partial rows meet in log2(epl) xor shuffles, a fixed order).  epl > 1 is chosen for narrow
  // rows on few nodes, where tpr lanes per row leave most of the chip without a wave (32-float rows at N = 3588: 448
  // waves for 1024 SIMDs, every wave a serial walk over ~10 in-edges).
  using T = typename VecT<V>::type;
  const int group = tpr * epl;
  int rpb = blockDim.x / group;
  int i = blockIdx.x * rpb + threadIdx.x / group;
  int lane = threadIdx.x % tpr, el = (threadIdx.x % group) / tpr;
  const bool live = i < N;
  int s0 = 0, s1 = 0;
  if (live) { s0 = rowptr[i]; s1 = rowptr[i + 1]; }
  float scale = 1.f;
  if (mean != 0.f) scale = 1.f / (float)max(s1 - s0, 1);
  const T* R = reinterpret_cast<const T*>(rows);
  T* O = reinterpret_cast<T*>(out);
  for (int c = lane; c < cols; c += tpr) {
    T acc = vzero<V>();
    int s = s0 + el;
    for (; s + 3 * epl < s1; s += 4 * epl) {      // four rows in flight (no index -> row chain per edge)
      int e0 = perm ? perm[s] : s, e1 = perm ? perm[s + epl] : s + epl;
      int e2 = perm ? perm[s + 2 * epl] : s + 2 * epl, e3 = perm ? perm[s + 3 * epl] : s + 3 * epl;
      T v0 = R[(size_t)e0 * ldi_cols + c], v1 = R[(size_t)e1 * ldi_cols + c];
      T v2 = R[(size_t)e2 * ldi_cols + c], v3 = R[(size_t)e3 * ldi_cols + c];
      acc = vadd(vadd(vadd(vadd(acc, v0), v1), v2), v3);
    }
    for (; s < s1; s += epl) {
      int e = perm ? perm[s] : s;
      acc = vadd(acc, R[(size_t)e * ldi_cols + c]);
    }
    if (rowptr2 != nullptr && live) {            // a second CSR view of the same rows summed into the same output row
      const int q1 = rowptr2[i + 1];             // (gradient of x[src] + x[dst]: by-source and by-target segments)
      for (int q = rowptr2[i] + el; q < q1; q += epl) {
        int e = perm2 ? perm2[q] : q;
        acc = vadd(acc, R[(size_t)e * ldi_cols + c]);
      }
    }
    for (int o = tpr; o < group; o <<= 1) {
      float* a = reinterpret_cast<float*>(&acc);
#pragma unroll
      for (int k = 0; k < V; ++k) a[k] += __shfl_xor(a[k], o);
    }
    if (live && el == 0) O[(size_t)i * ldo_cols + c] = vscale(acc, scale);
  }
}

static inline int seg_epl(int N, int tpr) {     // edge lanes: fill the chip when N * tpr threads would not
  const int force = 0;
  if (force) return force * tpr <= 64 ? force : 1;
  const int wide = 0;
  if (tpr >= 32 && !wide) return 1;     // rows of >= 128 floats: half a wave per row already; edge lanes measured slower
  int epl = 1;
  while (epl < 4 && tpr * epl * 2 <= 64 && (long)N * tpr * epl < 256L * 1024) epl *= 2;
  return epl;
}

extern "C" int msde_segment_sum_rows2(const float* rows, int ldi, const int* rowptr, const int* perm, const int* rowptr2,
                                      const int* perm2, int N, int D, float scale_by_inv_count, float* out, int ldo,
                                      void* stream) {
  if (N < 0 || D <= 0 || !rowptr || !out) return MSDE_EINVAL;
  if (ldo <= 0) ldo = D;
  if (ldi <= 0) ldi = D;
  if (ldo < D || ldi < D) return MSDE_EINVAL;
  if (N == 0) return 0;
  const bool vec = D % 4 == 0 && ldo % 4 == 0 && ldi % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (vec) {
    int cols = D / 4, tpr = pick_tpr(cols), epl = seg_epl(N, tpr), rpb = 256 / (tpr * epl);
    MSDE_LAUNCH(segment_sum_rows_kernel<4>, dim3((N + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), rows,
                       rowptr, perm, rowptr2, perm2, N, cols, ldi / 4, tpr, epl, scale_by_inv_count, out, ldo / 4);
  } else {
    int cols = D, tpr = pick_tpr(cols), epl = seg_epl(N, tpr), rpb = 256 / (tpr * epl);
    MSDE_LAUNCH(segment_sum_rows_kernel<1>, dim3((N + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), rows,
                       rowptr, perm, rowptr2, perm2, N, cols, ldi, tpr, epl, scale_by_inv_count, out, ldo);
  }
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_segment_sum_rows(const float* rows, int ldi, const int* rowptr, const int* perm, int N, int D,
                                     float scale_by_inv_count, float* out, int ldo, void* stream) {
  return msde_segment_sum_rows2(rows, ldi, rowptr, perm, nullptr, nullptr, N, D, scale_by_inv_count, out, ldo, stream);
}

template <int V>
__global__ void pair_gather_add_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                       const int* __restrict__ src, const int* __restrict__ dst, int E, int cols,
                                       int ld_cols, int tpr, float* __restrict__ out) {
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
  const T* a = reinterpret_cast<const T*>(A) + (size_t)j * ld_cols;
  if (B) {
    const T* b = reinterpret_cast<const T*>(B) + (size_t)i * ld_cols;
    for (int c = lane; c < cols; c += tpr) O[c] = vadd(a[c], b[c]);
  } else {
    for (int c = lane; c < cols; c += tpr) O[c] = a[c];
  }
}

extern "C" int msde_pair_gather_add(const float* A, const float* B, int ld, const int* src, const int* dst, int E,
                                    int D, float* out, void* stream) {
  if (E < 0 || D <= 0 || !A || !B || !src || !dst || !out) return MSDE_EINVAL;
  if (ld == 0) ld = D;
  if (ld < D) return MSDE_EINVAL;
  if (E == 0) return 0;
  const bool vec = D % 4 == 0 && ld % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (vec) {
    int cols = D / 4, tpr = pick_tpr(cols), rpb = 256 / tpr;
    MSDE_LAUNCH(pair_gather_add_kernel<4>, dim3((E + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), A, B,
                       src, dst, E, cols, ld / 4, tpr, out);
  } else {
    int cols = D, tpr = pick_tpr(cols), rpb = 256 / tpr;
    MSDE_LAUNCH(pair_gather_add_kernel<1>, dim3((E + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), A, B,
                       src, dst, E, cols, ld, tpr, out);
  }
  MSDE_CHECK_LAUNCH();
  return 0;
}

#define MSDE_PAIR_STRIP 128      // edges per statistics strip of the fused edge_2D_emb kernels (hip.PAIR_STRIP must agree)
// ---- edge_2D_emb of the 2D->3D model (SDE_model_2D_to_3D.py:35-40,264-271: Linear(cat(h_row, h_col)) -> BatchNorm1d -> ReLU ->
// Linear), fused around the gather (hip._PairBnReluLinear):
//   forward   pre[e] = A[src_e] + B[dst_e] written ONCE together with its per-strip BatchNorm statistics (this kernel; the
//             separate statistics pass re-read the 42 MB tensor), BatchNorm apply + ReLU in the A load of the second Linear;
//   backward  the BatchNorm input gradient dz = p g + w z + u is linear in (g, z), so the two segment sums over a node's
//             edges are taken of g and z themselves and combined per column afterwards (msde_segment_sum_rows_bn): dz is
//             never materialised.
// One workgroup per strip of SR edges: CL = D / 4 column lanes x RL = 512 / CL row lanes (450 of 512 threads at D = 300).
// Statistics per strip and column in the MSDE_RS_STATS_BNFWD format (mean, sum of squared deviations) over the VALID edges
// (e < *e_valid, or E), accumulated around the strip's first row (shifted sums: no cancellation for columns far from zero).
__global__ void __launch_bounds__(512)
pair_gather_add_stats_kernel(const float4* __restrict__ A, const float4* __restrict__ B, int ld4, const int* __restrict__ src,
                             const int* __restrict__ dst, int E, const int* __restrict__ e_valid, int CL, int SR,
                             float4* __restrict__ out, float* __restrict__ stats) {
  __shared__ float4 p1[512], p2[512];
  const int RL = 512 / CL;
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int e0 = blockIdx.x * SR;
  const int ev = e_valid ? min(E, e_valid[0]) : E;
  const bool active = rl < RL;
  float4 sh = make_float4(0.f, 0.f, 0.f, 0.f), s1 = sh, s2 = sh;
  if (active && e0 < ev) {
    const int j = src[e0], i = dst[e0];
    if (j >= 0) sh = vadd(A[(size_t)j * ld4 + cl], B[(size_t)i * ld4 + cl]);
  }
  if (active) {
    for (int r = rl; r < SR; r += 4 * RL) {        // four rows in flight per lane
      float4 v[4];
      int e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        e[k] = e0 + r + k * RL;
        const bool in = (r + k * RL < SR) && e[k] < E;
        const int j = in ? src[e[k]] : -1, i = in ? dst[e[k]] : 0;
        v[k] = j >= 0 ? vadd(A[(size_t)j * ld4 + cl], B[(size_t)i * ld4 + cl]) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (!in) e[k] = -1;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (e[k] < 0) continue;
        out[(size_t)e[k] * CL + cl] = v[k];
        if (e[k] < ev) {
          const float4 d = make_float4(v[k].x - sh.x, v[k].y - sh.y, v[k].z - sh.z, v[k].w - sh.w);
          s1 = vadd(s1, d);
          s2.x = fmaf(d.x, d.x, s2.x); s2.y = fmaf(d.y, d.y, s2.y); s2.z = fmaf(d.z, d.z, s2.z); s2.w = fmaf(d.w, d.w, s2.w);
        }
      }
    }
  }
  p1[threadIdx.x] = s1;
  p2[threadIdx.x] = s2;
  __syncthreads();
  if (rl == 0) {
    for (int q = 1; q < RL; ++q) { s1 = vadd(s1, p1[q * CL + cl]); s2 = vadd(s2, p2[q * CL + cl]); }     // row lanes in order
    const int n = max(0, min(SR, ev - e0));
    const float inv = n > 0 ? 1.f / (float)n : 0.f;
    float4 mean, m2;
    mean.x = sh.x + s1.x * inv; mean.y = sh.y + s1.y * inv; mean.z = sh.z + s1.z * inv; mean.w = sh.w + s1.w * inv;
    m2.x = fmaxf(s2.x - s1.x * s1.x * inv, 0.f); m2.y = fmaxf(s2.y - s1.y * s1.y * inv, 0.f);
    m2.z = fmaxf(s2.z - s1.z * s1.z * inv, 0.f); m2.w = fmaxf(s2.w - s1.w * s1.w * inv, 0.f);
    float4* st = reinterpret_cast<float4*>(stats + (size_t)blockIdx.x * 2 * (4 * CL));
    st[cl] = mean;
    st[CL + cl] = m2;
  }
}

// out[e] = A[src_e] + B[dst_e] (rows of width D, row stride ld of A / B) and stats[strip][2][D] over strips of
// MSDE_PAIR_STRIP edges (msde_bn_fin_fwd with that strip_rows).  D % 4 == 0, D <= 1024, 16-byte aligned operands.
extern "C" int msde_pair_gather_add_stats(const float* A, const float* B, int ld, const int* src, const int* dst, int E,
                                          int D, const int* e_valid, float* out, float* stats, void* stream) {
  if (E < 0 || D <= 0 || !A || !B || !src || !dst || !out || !stats) return MSDE_EINVAL;
  if (ld == 0) ld = D;
  if (ld < D) return MSDE_EINVAL;
  if (D % 4 || ld % 4 || D > 1024 ||
      ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(out) |
        reinterpret_cast<uintptr_t>(stats)) & 15))
    return MSDE_EUNSUP;
  if (E == 0) return 0;
  MSDE_LAUNCH(pair_gather_add_stats_kernel, dim3((E + MSDE_PAIR_STRIP - 1) / MSDE_PAIR_STRIP), dim3(512), 0, as_stream(stream),
              reinterpret_cast<const float4*>(A), reinterpret_cast<const float4*>(B), ld / 4, src, dst, E, e_valid, D / 4,
              MSDE_PAIR_STRIP, reinterpret_cast<float4*>(out), stats);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// Backward half: GA[e] = (g[e] W) gated by the ReLU behind the BatchNorm (scale z + shift > 0, z = Z[e]: the sign the forward
// product's A load saw), written once, with the BatchNorm-backward strip sums (sum GA, sum GA (z - mean)) of MSDE_RS_STATS_BNBWD
// over strips of MSDE_PAIR_STRIP edges.  g [E, H] with H <= 32: the product is 32 multiply-adds per output, done on the
// vector units from a register copy of W's column quad (the 2-D tiled product spends its time in a 300-column epilogue
// when K = 32: 111 us against ~25 here).  Thread layout of pair_gather_add_stats_kernel.
template <int H>
__device__ __forceinline__ float4 pbd_row(const float* __restrict__ gr_, const float4 (&w)[H]) {
  const float4* gr = reinterpret_cast<const float4*>(gr_);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < H / 4; ++q) {
    const float4 gv = gr[q];
    a.x = fmaf(gv.x, w[4 * q].x, a.x); a.y = fmaf(gv.x, w[4 * q].y, a.y); a.z = fmaf(gv.x, w[4 * q].z, a.z); a.w = fmaf(gv.x, w[4 * q].w, a.w);
    a.x = fmaf(gv.y, w[4 * q + 1].x, a.x); a.y = fmaf(gv.y, w[4 * q + 1].y, a.y); a.z = fmaf(gv.y, w[4 * q + 1].z, a.z); a.w = fmaf(gv.y, w[4 * q + 1].w, a.w);
    a.x = fmaf(gv.z, w[4 * q + 2].x, a.x); a.y = fmaf(gv.z, w[4 * q + 2].y, a.y); a.z = fmaf(gv.z, w[4 * q + 2].z, a.z); a.w = fmaf(gv.z, w[4 * q + 2].w, a.w);
    a.x = fmaf(gv.w, w[4 * q + 3].x, a.x); a.y = fmaf(gv.w, w[4 * q + 3].y, a.y); a.z = fmaf(gv.w, w[4 * q + 3].z, a.z); a.w = fmaf(gv.w, w[4 * q + 3].w, a.w);
  }
  return a;
}

template <int H>
__global__ void __launch_bounds__(512)
pair_bn_dgrad_stats_kernel(const float* __restrict__ g, int ldg, const float4* __restrict__ W, const float4* __restrict__ Z,
                           const float4* __restrict__ scale, const float4* __restrict__ shift, const float4* __restrict__ mean,
                           int E, const int* __restrict__ e_valid, int CL, int SR, float4* __restrict__ GA,
                           float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float gs[];      // [SR][H] gradient rows of the strip
  __shared__ float4 p1[512], p2[512];
  const int RL = 512 / CL;
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int e0 = blockIdx.x * SR;
  const int ev = e_valid ? min(E, e_valid[0]) : E;
  const bool active = rl < RL;
  for (int t = threadIdx.x; t < SR * (H / 4); t += 512) {
    const int r = t / (H / 4), q = t - r * (H / 4);
    const int e = e0 + r;
    reinterpret_cast<float4*>(gs)[t] = e < E ? *reinterpret_cast<const float4*>(g + (size_t)e * ldg + 4 * q)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 w[H];
  float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sf = sc, mu = sc;
  if (active) {
#pragma unroll
    for (int h = 0; h < H; ++h) w[h] = W[(size_t)h * CL + cl];
    sc = scale[cl]; sf = shift[cl]; mu = mean[cl];
  }
  __syncthreads();
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  auto finish = [&](int e, float4 a, const float4& z) __attribute__((always_inline)) {
    a.x = fmaf(z.x, sc.x, sf.x) > 0.f ? a.x : 0.f;
    a.y = fmaf(z.y, sc.y, sf.y) > 0.f ? a.y : 0.f;
    a.z = fmaf(z.z, sc.z, sf.z) > 0.f ? a.z : 0.f;
    a.w = fmaf(z.w, sc.w, sf.w) > 0.f ? a.w : 0.f;
    GA[(size_t)e * CL + cl] = a;
    if (e < ev) {
      s1 = vadd(s1, a);
      s2.x = fmaf(a.x, z.x - mu.x, s2.x); s2.y = fmaf(a.y, z.y - mu.y, s2.y);
      s2.z = fmaf(a.z, z.z - mu.z, s2.z); s2.w = fmaf(a.w, z.w - mu.w, s2.w);
    }
  };
  if (active) {
    // the z rows of this lane's next FOUR edges are requested before the 4 x 32 x 4 multiply-adds of the current four
    const int nrows = min(SR, E - e0);
    for (int r = rl; r < nrows; r += 4 * RL) {
      float4 z[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rr = r + k * RL;
        z[k] = rr < nrows ? Z[(size_t)(e0 + rr) * CL + cl] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rr = r + k * RL;
        if (rr < nrows) finish(e0 + rr, pbd_row<H>(gs + rr * H, w), z[k]);
      }
    }
  }
  p1[threadIdx.x] = s1;
  p2[threadIdx.x] = s2;
  __syncthreads();
  if (rl == 0) {
    for (int q = 1; q < RL; ++q) { s1 = vadd(s1, p1[q * CL + cl]); s2 = vadd(s2, p2[q * CL + cl]); }
    float4* st = reinterpret_cast<float4*>(stats + (size_t)blockIdx.x * 2 * (4 * CL));
    st[cl] = s1;
    st[CL + cl] = s2;
  }
}

extern "C" int msde_pair_bn_dgrad_stats(const float* g, int ldg, const float* W, const float* Z, const float* scale,
                                        const float* shift, const float* mean, int E, int H, int D, const int* e_valid,
                                        float* GA, float* stats, void* stream) {
  if (E < 0 || H <= 0 || D <= 0 || !g || !W || !Z || !scale || !shift || !mean || !GA || !stats) return MSDE_EINVAL;
  if (ldg <= 0) ldg = H;
  if ((H != 16 && H != 32) || D % 4 || D > 1024 || ldg % 4 ||
      ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(Z) |
        reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(mean) |
        reinterpret_cast<uintptr_t>(GA) | reinterpret_cast<uintptr_t>(stats)) & 15))
    return MSDE_EUNSUP;
  if (E == 0) return 0;
  const dim3 grid((E + MSDE_PAIR_STRIP - 1) / MSDE_PAIR_STRIP);
  const size_t lds = (size_t)MSDE_PAIR_STRIP * H * sizeof(float);
#define PBD_GO(H_) MSDE_LAUNCH(pair_bn_dgrad_stats_kernel<H_>, grid, dim3(512), lds, as_stream(stream), g, ldg,                     \
                               reinterpret_cast<const float4*>(W), reinterpret_cast<const float4*>(Z),                             \
                               reinterpret_cast<const float4*>(scale), reinterpret_cast<const float4*>(shift),                    \
                               reinterpret_cast<const float4*>(mean), E, e_valid, D / 4, MSDE_PAIR_STRIP,                          \
                               reinterpret_cast<float4*>(GA), stats)
  if (H == 16) PBD_GO(16); else PBD_GO(32);
#undef PBD_GO
  MSDE_CHECK_LAUNCH();
  return 0;
}

// Gradient of AB = [A | B] [N, 2 D] for pre[e] = A[src_e] + B[dst_e] through the BatchNorm input gradient dz = p GA + w z + u
// (z = pre), ONE launch over 2 N rows: row j < N is node j's by-SOURCE segment (gradient of A[j]), row N + i node i's
// by-TARGET segment (gradient of B[i]).  sum_e z[e] over a segment is taken from AB itself -- n A[j] + sum B[dst_e], or
// sum A[src_e] + n B[i] -- whose rows stay in L2, instead of from the [E, D] tensor: only GA is streamed from memory.
__global__ void __launch_bounds__(256)
pair_bn_scatter_kernel(const float4* __restrict__ GA, const float4* __restrict__ AB, int ldab4,
                       const int* __restrict__ src, const int* __restrict__ dst,
                       const int* __restrict__ rowptr_s, const int* __restrict__ perm_s,
                       const int* __restrict__ rowptr, int N, int cols, const float4* __restrict__ p,
                       const float4* __restrict__ w, const float4* __restrict__ u, float4* __restrict__ out, int ldo4) {
  // one column quad per thread, 256 / cols rows per workgroup (3 x 75 = 225 of 256 threads at D = 300)
  const int rpb = 256 / cols;
  const int c = threadIdx.x % cols, rw = threadIdx.x / cols;
  const int row = blockIdx.x * rpb + rw;
  if (rw >= rpb || row >= 2 * N) return;
  const bool by_src = row < N;
  const int i = by_src ? row : row - N;
  const int* rp = by_src ? rowptr_s : rowptr;
  const int s0 = rp[i], s1 = rp[i + 1];
  const float cnt = (float)(s1 - s0);
  const float4* other = AB + (by_src ? cols : 0);                         // B[.] / A[.] of the edge's other end
  const int* oidx = by_src ? dst : src;
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), az = ag;
  int s = s0;
  for (; s + 3 < s1; s += 4) {
    int e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = by_src ? perm_s[s + k] : s + k;
    float4 gv[4], zv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gv[k] = GA[(size_t)e[k] * cols + c];
      zv[k] = other[(size_t)oidx[e[k]] * ldab4 + c];
    }
    ag = vadd(vadd(vadd(vadd(ag, gv[0]), gv[1]), gv[2]), gv[3]);
    az = vadd(vadd(vadd(vadd(az, zv[0]), zv[1]), zv[2]), zv[3]);
  }
  for (; s < s1; ++s) {
    const int e = by_src ? perm_s[s] : s;
    ag = vadd(ag, GA[(size_t)e * cols + c]);
    az = vadd(az, other[(size_t)oidx[e] * ldab4 + c]);
  }
  const float4 o = AB[(size_t)i * ldab4 + (by_src ? 0 : cols) + c], pv = p[c], wv = w[c], uv = u[c];
  float4 r;
  r.x = fmaf(pv.x, ag.x, fmaf(wv.x, fmaf(cnt, o.x, az.x), cnt * uv.x));
  r.y = fmaf(pv.y, ag.y, fmaf(wv.y, fmaf(cnt, o.y, az.y), cnt * uv.y));
  r.z = fmaf(pv.z, ag.z, fmaf(wv.z, fmaf(cnt, o.z, az.z), cnt * uv.z));
  r.w = fmaf(pv.w, ag.w, fmaf(wv.w, fmaf(cnt, o.w, az.w), cnt * uv.w));
  out[(size_t)i * ldo4 + (by_src ? 0 : cols) + c] = r;
}

extern "C" int msde_pair_bn_scatter(const float* GA, const float* AB, int D, const int* src, const int* dst,
                                    const int* rowptr_s, const int* perm_s, const int* rowptr, int N, const float* p,
                                    const float* w, const float* u, float* gAB, void* stream) {
  if (N < 0 || D <= 0 || !GA || !AB || !src || !dst || !rowptr_s || !perm_s || !rowptr || !p || !w || !u || !gAB)
    return MSDE_EINVAL;
  if (D % 4 || ((reinterpret_cast<uintptr_t>(GA) | reinterpret_cast<uintptr_t>(AB) | reinterpret_cast<uintptr_t>(gAB) |
                 reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(u)) & 15))
    return MSDE_EUNSUP;
  if (N == 0) return 0;
  if (D > 1024) return MSDE_EUNSUP;
  const int cols = D / 4, rpb = 256 / cols;
  MSDE_LAUNCH(pair_bn_scatter_kernel, dim3((2 * N + rpb - 1) / rpb), dim3(256), 0, as_stream(stream),
              reinterpret_cast<const float4*>(GA), reinterpret_cast<const float4*>(AB), 2 * cols, src, dst, rowptr_s, perm_s,
              rowptr, N, cols, reinterpret_cast<const float4*>(p), reinterpret_cast<const float4*>(w),
              reinterpret_cast<const float4*>(u), reinterpret_cast<float4*>(gAB), 2 * cols);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// out[i] = p * (sum_e G[e]) + w * (sum_e Z[e]) + count_i * u over the edges e of node i's CSR segment (perm: edge ids, or
// the segment itself): the segment sum of the BatchNorm input gradient p G + w Z + u (vectors of msde_bn_fin_bwd) without
// forming it.  Thread layout of segment_sum_rows_kernel (tpr column lanes per row, float4 columns).
__global__ void segment_sum_rows_bn_kernel(const float4* __restrict__ G, const float4* __restrict__ Z, int ld4,
                                           const int* __restrict__ rowptr, const int* __restrict__ perm, int N, int cols,
                                           int tpr, const float4* __restrict__ p, const float4* __restrict__ w,
                                           const float4* __restrict__ u, float4* __restrict__ out, int ldo4) {
  const int rpb = blockDim.x / tpr;
  const int i = blockIdx.x * rpb + threadIdx.x / tpr, lane = threadIdx.x % tpr;
  if (i >= N) return;
  const int s0 = rowptr[i], s1 = rowptr[i + 1];
  const float cnt = (float)(s1 - s0);
  for (int c = lane; c < cols; c += tpr) {
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), az = ag;
    int s = s0;
    for (; s + 3 < s1; s += 4) {
      const int e0 = perm ? perm[s] : s, e1 = perm ? perm[s + 1] : s + 1, e2 = perm ? perm[s + 2] : s + 2,
                e3 = perm ? perm[s + 3] : s + 3;
      const float4 g0 = G[(size_t)e0 * ld4 + c], g1 = G[(size_t)e1 * ld4 + c], g2 = G[(size_t)e2 * ld4 + c],
                   g3 = G[(size_t)e3 * ld4 + c];
      const float4 z0 = Z[(size_t)e0 * ld4 + c], z1 = Z[(size_t)e1 * ld4 + c], z2 = Z[(size_t)e2 * ld4 + c],
                   z3 = Z[(size_t)e3 * ld4 + c];
      ag = vadd(vadd(vadd(vadd(ag, g0), g1), g2), g3);
      az = vadd(vadd(vadd(vadd(az, z0), z1), z2), z3);
    }
    for (; s < s1; ++s) {
      const int e = perm ? perm[s] : s;
      ag = vadd(ag, G[(size_t)e * ld4 + c]);
      az = vadd(az, Z[(size_t)e * ld4 + c]);
    }
    const float4 pv = p[c], wv = w[c], uv = u[c];
    float4 r;
    r.x = fmaf(pv.x, ag.x, fmaf(wv.x, az.x, cnt * uv.x));
    r.y = fmaf(pv.y, ag.y, fmaf(wv.y, az.y, cnt * uv.y));
    r.z = fmaf(pv.z, ag.z, fmaf(wv.z, az.z, cnt * uv.z));
    r.w = fmaf(pv.w, ag.w, fmaf(wv.w, az.w, cnt * uv.w));
    out[(size_t)i * ldo4 + c] = r;
  }
}

extern "C" int msde_segment_sum_rows_bn(const float* G, const float* Z, int ld, const int* rowptr, const int* perm, int N,
                                        int D, const float* p, const float* w, const float* u, float* out, int ldo,
                                        void* stream) {
  if (N < 0 || D <= 0 || !G || !Z || !rowptr || !p || !w || !u || !out) return MSDE_EINVAL;
  if (ld <= 0) ld = D;
  if (ldo <= 0) ldo = D;
  if (ld < D || ldo < D) return MSDE_EINVAL;
  if (D % 4 || ld % 4 || ldo % 4 ||
      ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(out) |
        reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(u)) & 15))
    return MSDE_EUNSUP;
  if (N == 0) return 0;
  const int cols = D / 4, tpr = pick_tpr(cols), rpb = 256 / tpr;
  MSDE_LAUNCH(segment_sum_rows_bn_kernel, dim3((N + rpb - 1) / rpb), dim3(256), 0, as_stream(stream),
              reinterpret_cast<const float4*>(G), reinterpret_cast<const float4*>(Z), ld / 4, rowptr, perm, N, cols, tpr,
              reinterpret_cast<const float4*>(p), reinterpret_cast<const float4*>(w), reinterpret_cast<const float4*>(u),
              reinterpret_cast<float4*>(out), ldo / 4);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// out[e] = [X[src_e] + X[dst_e] | C[e]]: the concatenation the basis MLP of the 2D->3D score network reads
// (equivariant_scorenetwork.py: cat([h_row + h_col, edge_attr])) written by the gather itself.  D, D2 in float4 units.
__global__ void pair_gather_cat_kernel(const float4* __restrict__ X, int ldx, const float4* __restrict__ C, int ldc,
                                       const int* __restrict__ src, const int* __restrict__ dst, int E, int D, int D2,
                                       int tpr, float4* __restrict__ out) {
  const int rpb = blockDim.x / tpr;
  const int e = blockIdx.x * rpb + threadIdx.x / tpr;
  const int lane = threadIdx.x % tpr;
  if (e >= E) return;
  const int j = src[e], i = dst[e];
  float4* O = out + (size_t)e * (D + D2);
  if (j < 0) {
    for (int c = lane; c < D + D2; c += tpr) O[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float4* a = X + (size_t)j * ldx;
  const float4* b = X + (size_t)i * ldx;
  const float4* cc = C + (size_t)e * ldc;
  for (int c = lane; c < D + D2; c += tpr) {
    float4 v;
    if (c < D) {
      const float4 p = a[c], q = b[c];
      v = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
    } else {
      v = cc[c - D];
    }
    O[c] = v;
  }
}

extern "C" int msde_pair_gather_cat(const float* X, int ldx, const float* C, int ldc, const int* src, const int* dst,
                                    int E, int D, int D2, float* out, void* stream) {
  if (E < 0 || D <= 0 || D2 <= 0 || !X || !C || !src || !dst || !out || ldx < D || ldc < D2) return MSDE_EINVAL;
  if ((D | D2 | ldx | ldc) & 3) return MSDE_EUNSUP;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(out)) & 15) return MSDE_EUNSUP;
  if (E == 0) return 0;
  const int cols = (D + D2) / 4, tpr = pick_tpr(cols), rpb = 256 / tpr;
  MSDE_LAUNCH(pair_gather_cat_kernel, dim3((E + rpb - 1) / rpb), dim3(256), 0, as_stream(stream),
              reinterpret_cast<const float4*>(X), ldx / 4, reinterpret_cast<const float4*>(C), ldc / 4, src, dst, E, D / 4,
              D2 / 4, tpr, reinterpret_cast<float4*>(out));
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_gather_rows(const float* X, const int* idx, int E, int D, float* out, void* stream) {
  if (E < 0 || D <= 0 || !X || !idx || !out) return MSDE_EINVAL;
  if (E == 0) return 0;
  if (D % 4 == 0) {
    int cols = D / 4, tpr = pick_tpr(cols), rpb = 256 / tpr;
    MSDE_LAUNCH(pair_gather_add_kernel<4>, dim3((E + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), X,
                       (const float*)nullptr, idx, (const int*)nullptr, E, cols, cols, tpr, out);
  } else {
    int cols = D, tpr = pick_tpr(cols), rpb = 256 / tpr;
    MSDE_LAUNCH(pair_gather_add_kernel<1>, dim3((E + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), X,
                       (const float*)nullptr, idx, (const int*)nullptr, E, cols, cols, tpr, out);
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
    // same left-to-right order as the reference's  x_embedding += emb_k(x[:,k])  (starts from 0); codes first,
    // then all table rows in flight (no code -> row chain per feature)
    if (K <= 16) {
      int cd[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) cd[k] = k < K ? codes[(size_t)i * K + k] : 0;
      T v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = k < K ? Tb[(size_t)cd[k] * cols + c] : vzero<V>();
      T acc = v[0];
#pragma unroll
      for (int k = 1; k < 16; ++k)
        if (k < K) acc = vadd(acc, v[k]);
      O[c] = acc;
    } else {
      T acc = Tb[(size_t)codes[(size_t)i * K] * cols + c];
      for (int k = 1; k < K; ++k) acc = vadd(acc, Tb[(size_t)codes[(size_t)i * K + k] * cols + c]);
      O[c] = acc;
    }
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

// Table gradient g_tab[r] = sum over list(r) of g[node], deterministic and atomic-free:
//   pass 1, block (r, s, column tile): slice s of row r's node list, 4 item lanes x 64 column lanes, four row
//           loads in flight per thread; the 4 item lanes are combined in lane order and the partial row goes to
//           slab[s][r][:].  Row r is cut into slices(r) = clamp(ceil(len/32), 1, split) slices.
//   pass 2, one thread per table element: sums its row's slices in index order (zero for an empty list).
#define ES_ITEMS_PER_SLICE 32
__device__ __forceinline__ int es_slices(int len, int split) {
  int s = (len + ES_ITEMS_PER_SLICE - 1) / ES_ITEMS_PER_SLICE;
  return s < 1 ? 1 : (s > split ? split : s);
}

template <int V>
__global__ void __launch_bounds__(256)
embedding_sum_bwd_partial_kernel(const float* __restrict__ g, const int* __restrict__ list_ptr,
                                 const int* __restrict__ list_nodes, int R, int D, int split, float* __restrict__ slab) {
  using T = typename VecT<V>::type;
  __shared__ T part[4][64];
  const int r = blockIdx.x, s = blockIdx.y;
  const int p0 = list_ptr[r], len = list_ptr[r + 1] - p0;
  const int ns = es_slices(len, split);
  if (len == 0 || s >= ns) return;
  const int chunk = (len + ns - 1) / ns;
  const int a = p0 + s * chunk, b = min(a + chunk, p0 + len);
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const int c = (blockIdx.z * 64 + lx) * V;
  T acc = vzero<V>(), acc2 = vzero<V>();
  if (c < D) {
    // 16 items per pass and item lane: all node ids first, then all 16 row loads in flight (a dependent
    // id -> row chain per item would make the block latency bound), summed in list order
    for (int p0 = a + ly; p0 < b; p0 += 64) {
      int ids[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        int p = p0 + 4 * k;
        ids[k] = p < b ? list_nodes[p] : -1;
      }
      T v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k)
        v[k] = ids[k] >= 0 ? *reinterpret_cast<const T*>(g + (size_t)ids[k] * D + c) : vzero<V>();
#pragma unroll
      for (int k = 0; k < 16; k += 2) { acc = vadd(acc, v[k]); acc2 = vadd(acc2, v[k + 1]); }
    }
    acc = vadd(acc, acc2);
  }
  part[ly][lx] = acc;
  __syncthreads();
  if (ly == 0 && c < D) {
    T t = vadd(vadd(vadd(part[0][lx], part[1][lx]), part[2][lx]), part[3][lx]);
    *reinterpret_cast<T*>(slab + ((size_t)s * R + r) * D + c) = t;
  }
}

__global__ void __launch_bounds__(256)
embedding_sum_bwd_reduce_kernel(const float* __restrict__ slab, const int* __restrict__ list_ptr, int R, int D, int split,
                                float* __restrict__ g_tab) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)R * D) return;
  int r = (int)(i / D);
  int len = list_ptr[r + 1] - list_ptr[r];
  float acc = 0.f;
  if (len > 0) {
    const int ns = es_slices(len, split);
    const size_t st = (size_t)R * D;
    int s = 0;
    for (; s + 8 <= ns; s += 8) {          // eight slices in flight, added in slice order
      float v0 = slab[(size_t)s * st + i], v1 = slab[(size_t)(s + 1) * st + i], v2 = slab[(size_t)(s + 2) * st + i];
      float v3 = slab[(size_t)(s + 3) * st + i], v4 = slab[(size_t)(s + 4) * st + i], v5 = slab[(size_t)(s + 5) * st + i];
      float v6 = slab[(size_t)(s + 6) * st + i], v7 = slab[(size_t)(s + 7) * st + i];
      acc = (((((((acc + v0) + v1) + v2) + v3) + v4) + v5) + v6) + v7;
    }
    for (; s < ns; ++s) acc += slab[(size_t)s * st + i];
  }
  g_tab[i] = acc;
}

extern "C" long long msde_embedding_sum_bwd_workspace_floats(int R, int D, int split) {
  if (split < 1) split = 1;
  return (long long)split * R * D;
}

extern "C" int msde_embedding_sum_bwd(const float* g, const int* list_ptr, const int* list_nodes, int R, int D,
                                      int split, float* g_tab, float* workspace, void* stream) {
  if (R <= 0 || D <= 0 || split <= 0 || !g || !list_ptr || !list_nodes || !g_tab || !workspace) return MSDE_EINVAL;
  hipStream_t st = as_stream(stream);
  const bool vec = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(g) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
  if (vec)
    MSDE_LAUNCH(embedding_sum_bwd_partial_kernel<4>, dim3(R, split, (D / 4 + 63) / 64), dim3(256), 0, st, g, list_ptr,
                list_nodes, R, D, split, workspace);
  else
    MSDE_LAUNCH(embedding_sum_bwd_partial_kernel<1>, dim3(R, split, (D + 63) / 64), dim3(256), 0, st, g, list_ptr,
                list_nodes, R, D, split, workspace);
  MSDE_CHECK_LAUNCH();
  size_t n = (size_t)R * D;
  MSDE_LAUNCH(embedding_sum_bwd_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
              (const float*)workspace, list_ptr, R, D, split, g_tab);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_abi_version(void) { return 1; }
extern "C" const char* msde_target_arch(void) { return "gfx950"; }


// ---- diagnostics: a device timestamp on a stream (tools/probes/step_timeline.py).  One thread stores the 100 MHz
// real-time counter; captured into the step's hipGraph it gives an undistorted timeline of the replay (rocprofv3's
// kernel trace slows every dispatch and serialises the two queues of the step).
__global__ void debug_stamp_kernel(long long* slot) { *slot = (long long)wall_clock64(); }
extern "C" int msde_debug_stamp(long long* slot, void* stream) {
  if (!slot) return MSDE_EINVAL;
  MSDE_LAUNCH(debug_stamp_kernel, dim3(1), dim3(1), 0, as_stream(stream), slot);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_pair_strip(void) { return MSDE_PAIR_STRIP; }
