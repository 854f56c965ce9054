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
    int e = s0;
    for (; e + 1 < s1; e += 2) {      // two edges (2 neighbour rows + 6 table rows) in flight, added in edge order
      int j0 = src[e], j1 = src[e + 1];
      int a0 = codes[3 * e], a1 = codes[3 * e + 1], a2 = codes[3 * e + 2];
      int b0 = codes[3 * e + 3], b1 = codes[3 * e + 4], b2 = codes[3 * e + 5];
      T x0 = X[(size_t)j0 * cols + c], x1 = X[(size_t)j1 * cols + c];
      T t0 = Tb[(size_t)a0 * cols + c], t1 = Tb[(size_t)a1 * cols + c], t2 = Tb[(size_t)a2 * cols + c];
      T u0 = Tb[(size_t)b0 * cols + c], u1 = Tb[(size_t)b1 * cols + c], u2 = Tb[(size_t)b2 * cols + c];
      acc = vadd(acc, vrelu(vadd(x0, vadd(vadd(t0, t1), t2))));
      acc = vadd(acc, vrelu(vadd(x1, vadd(vadd(u0, u1), u2))));
    }
    for (; e < s1; ++e) {
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
    int s = s0;
    for (; s + 1 < s1; s += 2) {      // two edges in flight (indices, then 2 gradient rows + 6 table rows)
      int e0 = perm_s[s], e1 = perm_s[s + 1];
      int d0 = dst[e0], d1 = dst[e1];
      int a0 = codes[3 * e0], a1 = codes[3 * e0 + 1], a2 = codes[3 * e0 + 2];
      int b0 = codes[3 * e1], b1 = codes[3 * e1 + 1], b2 = codes[3 * e1 + 2];
      T g0 = G[(size_t)d0 * cols + c], g1 = G[(size_t)d1 * cols + c];
      T t0 = Tb[(size_t)a0 * cols + c], t1 = Tb[(size_t)a1 * cols + c], t2 = Tb[(size_t)a2 * cols + c];
      T u0 = Tb[(size_t)b0 * cols + c], u1 = Tb[(size_t)b1 * cols + c], u2 = Tb[(size_t)b2 * cols + c];
      acc = vadd(acc, vgate(g0, vadd(xj, vadd(vadd(t0, t1), t2))));
      acc = vadd(acc, vgate(g1, vadd(xj, vadd(vadd(u0, u1), u2))));
    }
    for (; s < s1; ++s) {
      int e = perm_s[s];
      T m = vadd(xj, bond_emb<V>(Tb, codes, e, cols, c));
      acc = vadd(acc, vgate(G[(size_t)dst[e] * cols + c], m));
    }
    O[(size_t)j * cols + c] = acc;
  }
}

// ---- variants fused with the BatchNorm of the PREVIOUS layer (csrc/gemm_rs.hip; D % 4 == 0) ---------------------
// Forward: the layer input is h = max(z scale[c] + shift[c], 0 if relu) of the previous layer's second product z (its
// outer BatchNorm, molecule_gnn_model.py:176-182): applied on the fly to every gathered row and to the node's own row,
// which is also written out (h_out) -- the separate BatchNorm-apply launch of the layer disappears.
__global__ void __launch_bounds__(256)
gin_aggregate_bn_fwd_kernel(const float* __restrict__ z, const float* __restrict__ scale, const float* __restrict__ shift,
                            int relu, const float* __restrict__ tab, const int* __restrict__ codes,
                            const float* __restrict__ eps, const int* __restrict__ rowptr, const int* __restrict__ src,
                            int N, int cols, float* __restrict__ h_out, float* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= N) return;
  const float4* Z = reinterpret_cast<const float4*>(z);
  const float4* Tb = reinterpret_cast<const float4*>(tab);
  const float ope = 1.f + eps[0];
  const int s0 = rowptr[i], s1 = rowptr[i + 1];
  for (int c = lane; c < cols; c += 64) {
    const float4 sc = reinterpret_cast<const float4*>(scale)[c], sh = reinterpret_cast<const float4*>(shift)[c];
    auto hrow = [&](int j) {
      float4 v = vfma(Z[(size_t)j * cols + c], sc, sh);
      return relu ? vrelu(v) : v;
    };
    float4 acc = vzero4();
    int e = s0;
    for (; e + 1 < s1; e += 2) {
      const int j0 = src[e], j1 = src[e + 1];
      const int a0 = codes[3 * e], a1 = codes[3 * e + 1], a2 = codes[3 * e + 2];
      const int b0 = codes[3 * e + 3], b1 = codes[3 * e + 4], b2 = codes[3 * e + 5];
      const float4 x0 = hrow(j0), x1 = hrow(j1);
      const float4 t0 = Tb[(size_t)a0 * cols + c], t1 = Tb[(size_t)a1 * cols + c], t2 = Tb[(size_t)a2 * cols + c];
      const float4 u0 = Tb[(size_t)b0 * cols + c], u1 = Tb[(size_t)b1 * cols + c], u2 = Tb[(size_t)b2 * cols + c];
      acc = vadd(acc, vrelu(vadd(x0, vadd(vadd(t0, t1), t2))));
      acc = vadd(acc, vrelu(vadd(x1, vadd(vadd(u0, u1), u2))));
    }
    for (; e < s1; ++e) acc = vadd(acc, vrelu(vadd(hrow(src[e]), bond_emb<4>(Tb, codes, e, cols, c))));
    const float4 hi = hrow(i);
    reinterpret_cast<float4*>(h_out)[(size_t)i * cols + c] = hi;
    reinterpret_cast<float4*>(out)[(size_t)i * cols + c] = vadd(vscale(hi, ope), acc);
  }
}

// Backward w.r.t. the layer input + the BatchNorm-backward partial sums of that gradient for the previous layer's outer
// BatchNorm: per 16-row strip and column, sum g' and sum g' (z - mean[c]) with g' = g_x gated by x > 0 when `relu` (x is
// that BatchNorm's ReLU output) -- the [strips][2][D] format msde_bn_fin_bwd takes.  One workgroup = 16 rows.
__global__ void __launch_bounds__(256)
gin_aggregate_bwd_x_stats_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ tab,
                                 const int* __restrict__ codes, const float* __restrict__ eps,
                                 const int* __restrict__ rowptr_s, const int* __restrict__ perm_s,
                                 const int* __restrict__ dst, int N, const int* __restrict__ m_valid, int cols,
                                 const float* __restrict__ zprev, const float* __restrict__ mean, int relu,
                                 float* __restrict__ g_x, float* __restrict__ stats) {
  extern __shared__ float red[];                  // [3 groups][2][4 cols]
  const int grp = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int mv = m_valid ? min(N, m_valid[0]) : N;
  const float4* X = reinterpret_cast<const float4*>(x);
  const float4* G = reinterpret_cast<const float4*>(g);
  const float4* Tb = reinterpret_cast<const float4*>(tab);
  const float4* Zp = reinterpret_cast<const float4*>(zprev);
  const float ope = 1.f + eps[0];
  const int D = cols * 4;
  // (the trip count is the same for every lane -- the barriers below are reached by all threads; lanes past the last column
  // group only take part in them)
  for (int cb = 0; cb < cols; cb += 64) {
    const int c = cb + lane;
    const bool act = c < cols;
    const float4 mu = act ? reinterpret_cast<const float4*>(mean)[c] : vzero4();
    float4 sa = vzero4(), sb = vzero4();
    for (int it = 0; act && it < 4; ++it) {
      const int j = blockIdx.x * 16 + it * 4 + grp;
      if (j >= N) continue;
      const float4 xj = X[(size_t)j * cols + c];
      float4 acc = vscale(G[(size_t)j * cols + c], ope);
      const int s0 = rowptr_s[j], s1 = rowptr_s[j + 1];
      int sidx = s0;
      for (; sidx + 1 < s1; sidx += 2) {
        const int e0 = perm_s[sidx], e1 = perm_s[sidx + 1];
        const int d0 = dst[e0], d1 = dst[e1];
        const int a0 = codes[3 * e0], a1 = codes[3 * e0 + 1], a2 = codes[3 * e0 + 2];
        const int b0 = codes[3 * e1], b1 = codes[3 * e1 + 1], b2 = codes[3 * e1 + 2];
        const float4 g0 = G[(size_t)d0 * cols + c], g1 = G[(size_t)d1 * cols + c];
        const float4 t0 = Tb[(size_t)a0 * cols + c], t1 = Tb[(size_t)a1 * cols + c], t2 = Tb[(size_t)a2 * cols + c];
        const float4 u0 = Tb[(size_t)b0 * cols + c], u1 = Tb[(size_t)b1 * cols + c], u2 = Tb[(size_t)b2 * cols + c];
        acc = vadd(acc, vgate(g0, vadd(xj, vadd(vadd(t0, t1), t2))));
        acc = vadd(acc, vgate(g1, vadd(xj, vadd(vadd(u0, u1), u2))));
      }
      for (; sidx < s1; ++sidx) {
        const int e = perm_s[sidx];
        acc = vadd(acc, vgate(G[(size_t)dst[e] * cols + c], vadd(xj, bond_emb<4>(Tb, codes, e, cols, c))));
      }
      reinterpret_cast<float4*>(g_x)[(size_t)j * cols + c] = acc;
      if (j < mv) {
        const float4 gp = relu ? vgate(acc, xj) : acc;
        const float4 zz = Zp[(size_t)j * cols + c];
        sa = vadd(sa, gp);
        sb = vfma(gp, make_float4(zz.x - mu.x, zz.y - mu.y, zz.z - mu.z, zz.w - mu.w), sb);
      }
    }
    // groups 1..3 hand their sums to group 0 (fixed order)
    float* slot = red + ((size_t)(grp > 0 ? grp - 1 : 0) * 2) * D;
    if (grp > 0 && act) {
      reinterpret_cast<float4*>(slot)[c] = sa;
      reinterpret_cast<float4*>(slot + D)[c] = sb;
    }
    __syncthreads();
    if (grp == 0 && act) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        sa = vadd(sa, reinterpret_cast<const float4*>(red + (size_t)k * 2 * D)[c]);
        sb = vadd(sb, reinterpret_cast<const float4*>(red + (size_t)k * 2 * D + D)[c]);
      }
      reinterpret_cast<float4*>(stats + (size_t)blockIdx.x * 2 * D)[c] = sa;
      reinterpret_cast<float4*>(stats + (size_t)blockIdx.x * 2 * D + D)[c] = sb;
    }
    __syncthreads();
  }
}

extern "C" int msde_gin_aggregate_bn_fwd(const float* z, const float* scale, const float* shift, int relu, const float* tab,
                                         const int* codes, const float* eps, const int* rowptr, const int* src, int N,
                                         int D, float* h_out, float* out, void* stream) {
  if (N < 0 || D <= 0 || D % 4 || !z || !scale || !shift || !tab || !eps || !rowptr || !h_out || !out) return MSDE_EINVAL;
  if (N == 0) return 0;
  MSDE_LAUNCH(gin_aggregate_bn_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), z, scale, shift, relu, tab, codes,
              eps, rowptr, src, N, D / 4, h_out, out);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_gin_aggregate_bwd_x_stats(const float* g, const float* x, const float* tab, const int* codes,
                                              const float* eps, const int* rowptr_s, const int* perm_s, const int* dst,
                                              int N, const int* m_valid, int D, const float* zprev, const float* mean,
                                              int relu, float* g_x, float* stats, void* stream) {
  if (N < 0 || D <= 0 || D % 4 || !g || !x || !tab || !eps || !rowptr_s || !zprev || !mean || !g_x || !stats)
    return MSDE_EINVAL;
  if (N == 0) return 0;
  MSDE_LAUNCH(gin_aggregate_bwd_x_stats_kernel, dim3((N + 15) / 16), dim3(256), (size_t)3 * 2 * D * sizeof(float),
              as_stream(stream), g, x, tab, codes, eps, rowptr_s, perm_s, dst, N, m_valid, D / 4, zprev, mean, relu, g_x,
              stats);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// Bond-table and eps gradients, deterministic and free of global atomics.  Block b owns GT_EC consecutive
// edges (canonical by-target order) and a slice of the nodes; thread t owns column t (+ blockDim, ...) of the
// LDS copies of the table (for the ReLU gate x[src] + emb > 0) and of the block's partial gradient table, so
// the read-modify-writes never race.  Four edges' row loads are in flight per thread.  The partial tables
// (R*D floats per block) and eps partials are summed over blocks in index order by reduce_slabs.
static inline int gt_ec() {           // edges per workgroup (MSDE_GT_EC: tuning knob)
  const int v = 16;      // edges per workgroup (round 3: 8: 2.76, 16: 2.72, 32: 2.76, 64: 2.83 ms)
  return v;
}
#define GT_MAX_LAYERS 8
struct gt_layers {                  // the same kernel for up to GT_MAX_LAYERS layers of one graph in ONE launch
  const float* g[GT_MAX_LAYERS];
  const float* x[GT_MAX_LAYERS];
  const float* tab[GT_MAX_LAYERS];
  float* ws[GT_MAX_LAYERS];         // per layer: [nb][R*D] partial tables, then [nb] eps partials
};
__global__ void gin_aggregate_bwd_tab_kernel(const gt_layers L, int nb_per_layer, const int* __restrict__ codes,
                                             const int* __restrict__ src, const int* __restrict__ dst, int N, int E,
                                             const int* __restrict__ Ndev, const int* __restrict__ Edev,
                                             int D, int R, int nodes_per_block, int GT_EC) {
  const int layer = blockIdx.x / nb_per_layer, blk = blockIdx.x - layer * nb_per_layer;
  const float* __restrict__ g = L.g[layer];
  const float* __restrict__ x = L.x[layer];
  const float* __restrict__ tab = L.tab[layer];
  float* __restrict__ slabs = L.ws[layer];
  float* __restrict__ eps_part = L.ws[layer] + (size_t)nb_per_layer * R * D;
  N = msde_true_rows(N, Ndev);      // row bounds: padded edges / atoms contribute nothing
  E = msde_true_rows(E, Edev);
  extern __shared__ float lds[];  // [R*D] table copy, [R*D] partial gradient, [blockDim/64] reduction scratch
  float* stab = lds;
  float* ltab = lds + (size_t)R * D;
  float* red = ltab + (size_t)R * D;
  for (int t = threadIdx.x; t < R * D; t += blockDim.x) { stab[t] = tab[t]; ltab[t] = 0.f; }
  __syncthreads();
  const int e0 = blk * GT_EC, e1 = min(e0 + GT_EC, E);
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    for (int eb = e0; eb < e1; eb += 4) {
      float xv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int e = max(min(eb + u, e1 - 1), 0);
        xv[u] = x[(size_t)src[e] * D + c];
        gv[u] = g[(size_t)dst[e] * D + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int e = eb + u;
        if (e < e1) {
          int c0 = codes[3 * e], c1 = codes[3 * e + 1], c2 = codes[3 * e + 2];
          float emb = (stab[c0 * D + c] + stab[c1 * D + c]) + stab[c2 * D + c];
          if (xv[u] + emb > 0.f) {
            ltab[c0 * D + c] += gv[u];
            ltab[c1 * D + c] += gv[u];
            ltab[c2 * D + c] += gv[u];
          }
        }
      }
    }
  }
  // eps partial: sum_i g[i].x[i] over this block's node slice
  float eacc = 0.f;
  const int n0 = blk * nodes_per_block, n1 = min(n0 + nodes_per_block, N);
  for (int i = n0; i < n1; ++i)
    for (int c = threadIdx.x; c < D; c += blockDim.x) eacc = fmaf(g[(size_t)i * D + c], x[(size_t)i * D + c], eacc);
  __syncthreads();
  float* slab = slabs + (size_t)blk * R * D;
  for (int t = threadIdx.x; t < R * D; t += blockDim.x) slab[t] = ltab[t];
  eacc = group_sum(eacc, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = eacc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float sum = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += red[w];
    eps_part[blk] = sum;
  }
}

static inline int gt_blocks(int N, int E) {
  int nb = (E + gt_ec() - 1) / gt_ec();
  return nb < 1 ? 1 : nb;
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

extern "C" int msde_gin_aggregate_bwd_tab_slabs(int N, int E) { return gt_blocks(N, E); }

extern "C" long long msde_gin_aggregate_bwd_tab_workspace_floats(int N, int E, int D, int R) {
  return (long long)gt_blocks(N, E) * ((long long)R * D + 1);
}

static int gt_launch(const gt_layers& L, int layers, const int* codes, const int* src, const int* dst, int N, int E, int D,
                     int R, const int* n_dev, const int* e_dev, hipStream_t st) {
  int threads = ((D + 63) / 64) * 64;
  if (threads > 512) threads = 512;
  size_t lds = (2 * (size_t)R * D + 8) * sizeof(float);
  if (lds > 64 * 1024) return MSDE_EUNSUP;
  int nb = gt_blocks(N, E);
  int npb = (N + nb - 1) / nb;
  MSDE_LAUNCH(gin_aggregate_bwd_tab_kernel, dim3(nb * layers), dim3(threads), lds, st, L, nb, codes, src, dst, N, E,
              n_dev, e_dev, D, R, npb, gt_ec());
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_gin_aggregate_bwd_tab(const float* g, const float* x, const float* tab, const int* codes,
                                          const int* src, const int* dst, int N, int E, int D, int R, float* g_tab,
                                          float* g_eps, float* workspace, const int* n_dev, const int* e_dev,
                                          void* stream) {
  // g_tab == g_eps == NULL: leave the per-workgroup partial tables in `workspace` for a batched reduction
  const bool no_reduce = !g_tab && !g_eps;
  if (N < 0 || E < 0 || D <= 0 || R <= 0 || !g || !x || !tab || !workspace || (!no_reduce && (!g_tab || !g_eps)))
    return MSDE_EINVAL;
  if (E > 0 && (!codes || !src || !dst)) return MSDE_EINVAL;
  hipStream_t st = as_stream(stream);
  gt_layers L = {};
  L.g[0] = g; L.x[0] = x; L.tab[0] = tab; L.ws[0] = workspace;
  int rc = gt_launch(L, 1, codes, src, dst, N, E, D, R, n_dev, e_dev, st);
  if (rc) return rc;
  if (no_reduce) return 0;
  int nb = gt_blocks(N, E);
  return msde_reduce_slabs(workspace, nb, (size_t)R * D, g_tab, workspace + (size_t)nb * R * D, 1, g_eps, st);
}

// the bond-table / eps partials of `layers` <= 8 GIN layers over the SAME graph (codes, src, dst) in one launch: the
// layers' kernels are independent leaf work that the trainer runs beside the grouped weight-gradient launch, where five
// launches in a row each wait for their last workgroups to find a free CU.  g / x / tab / workspace: HOST arrays of
// `layers` device pointers; each workspace as for msde_gin_aggregate_bwd_tab with g_tab == g_eps == NULL.
extern "C" int msde_gin_aggregate_bwd_tab_multi(const float* const* g, const float* const* x, const float* const* tab,
                                                float* const* workspace, int layers, const int* codes, const int* src,
                                                const int* dst, int N, int E, int D, int R, const int* n_dev,
                                                const int* e_dev, void* stream) {
  if (layers < 1 || layers > GT_MAX_LAYERS || N < 0 || E < 0 || D <= 0 || R <= 0 || !g || !x || !tab || !workspace)
    return MSDE_EINVAL;
  if (E > 0 && (!codes || !src || !dst)) return MSDE_EINVAL;
  gt_layers L = {};
  for (int l = 0; l < layers; ++l) {
    if (!g[l] || !x[l] || !tab[l] || !workspace[l]) return MSDE_EINVAL;
    L.g[l] = g[l]; L.x[l] = x[l]; L.tab[l] = tab[l]; L.ws[l] = workspace[l];
  }
  return gt_launch(L, layers, codes, src, dst, N, E, D, R, n_dev, e_dev, as_stream(stream));
}
