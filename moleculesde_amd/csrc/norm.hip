// norm.hip — training-mode BatchNorm1d over rows (atoms or edges) with optional fused ReLU.
// molecule_gnn_model.py:17,159,176-182 (10 BatchNorms of the GIN stack) and
// SDE_model_2D_to_3D.py:265 (BatchNorm over ~35 k edges inside edge_2D_emb).
//
// Forward  = 2 launches: per-(row split, column) Welford partials, then combine (fixed order, Chan's
//            formula) + normalise + affine (+ ReLU) + running-stat update.
// Backward = 2 launches: per-split column sums of dz and dz*xhat, then combine + input gradient.
// Threads run along columns (64 consecutive floats per wave row = 256 B), 4 row lanes per block.
#include "msde_common.h"

#define BN_COLS 64
#define BN_RL 4   // row lanes per block

__global__ void __launch_bounds__(256)
bn_stats_partial_kernel(const float* __restrict__ X, int M, int C, int rows_per_split, float* __restrict__ ws) {
  __shared__ float s_n[BN_RL][BN_COLS], s_mean[BN_RL][BN_COLS], s_m2[BN_RL][BN_COLS];
  const int tx = threadIdx.x & (BN_COLS - 1), ty = threadIdx.x / BN_COLS;
  const int c = blockIdx.x * BN_COLS + tx;
  const int r0 = blockIdx.y * rows_per_split, r1 = min(r0 + rows_per_split, M);
  float n = 0.f, mean = 0.f, m2 = 0.f;
  if (c < C) {
    for (int r = r0 + ty; r < r1; r += BN_RL) {
      float x = X[(size_t)r * C + c];
      n += 1.f;
      float d = x - mean;
      mean += d / n;
      m2 = fmaf(d, x - mean, m2);
    }
  }
  s_n[ty][tx] = n; s_mean[ty][tx] = mean; s_m2[ty][tx] = m2;
  __syncthreads();
  if (ty == 0 && c < C) {
#pragma unroll
    for (int l = 1; l < BN_RL; ++l) {
      float nb = s_n[l][tx], mb = s_mean[l][tx], m2b = s_m2[l][tx];
      if (nb > 0.f) {
        float nn = n + nb, d = mb - mean;
        mean += d * (nb / nn);
        m2 += m2b + d * d * (n * nb / nn);
        n = nn;
      }
    }
    float* o = ws + ((size_t)blockIdx.y * C + c) * 3;
    o[0] = n; o[1] = mean; o[2] = m2;
  }
}

__device__ __forceinline__ void bn_combine(const float* __restrict__ ws, int splits, int C, int c, float& n, float& mean,
                                           float& m2) {
  n = 0.f; mean = 0.f; m2 = 0.f;
  for (int s = 0; s < splits; ++s) {
    const float* p = ws + ((size_t)s * C + c) * 3;
    float nb = p[0], mb = p[1], m2b = p[2];
    if (nb > 0.f) {
      float nn = n + nb, d = mb - mean;
      mean += d * (nb / nn);
      m2 += m2b + d * d * (n * nb / nn);
      n = nn;
    }
  }
}

__global__ void __launch_bounds__(256)
bn_fwd_apply_kernel(const float* __restrict__ X, const float* __restrict__ ws, int M, int C, int splits,
                    int rows_per_block, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                    float momentum, float* __restrict__ running_mean, float* __restrict__ running_var, int relu,
                    float* __restrict__ Y, float* __restrict__ save_mean, float* __restrict__ save_rstd) {
  __shared__ float s_scale[BN_COLS], s_shift[BN_COLS];
  __shared__ float s_n[BN_RL][BN_COLS], s_mean[BN_RL][BN_COLS], s_m2[BN_RL][BN_COLS];
  const int tx = threadIdx.x & (BN_COLS - 1), ty = threadIdx.x / BN_COLS;
  const int c = blockIdx.x * BN_COLS + tx;
  {  // each row lane merges the splits s = ty, ty+4, ...; the four partials are then merged in lane order
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (c < C) {
      for (int sidx = ty; sidx < splits; sidx += BN_RL) {
        const float* p = ws + ((size_t)sidx * C + c) * 3;
        float nb = p[0], mb = p[1], m2b = p[2];
        if (nb > 0.f) {
          float nn = n + nb, d = mb - mean;
          mean += d * (nb / nn);
          m2 += m2b + d * d * (n * nb / nn);
          n = nn;
        }
      }
    }
    s_n[ty][tx] = n; s_mean[ty][tx] = mean; s_m2[ty][tx] = m2;
  }
  __syncthreads();
  if (ty == 0 && c < C) {
    float n = s_n[0][tx], mean = s_mean[0][tx], m2 = s_m2[0][tx];
#pragma unroll
    for (int l = 1; l < BN_RL; ++l) {
      float nb = s_n[l][tx], mb = s_mean[l][tx], m2b = s_m2[l][tx];
      if (nb > 0.f) {
        float nn = n + nb, d = mb - mean;
        mean += d * (nb / nn);
        m2 += m2b + d * d * (n * nb / nn);
        n = nn;
      }
    }
    float var = m2 / n;
    float rstd = rsqrtf(var + eps);
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    s_scale[tx] = g * rstd;
    s_shift[tx] = b - mean * g * rstd;
    if (blockIdx.y == 0) {
      save_mean[c] = mean;
      save_rstd[c] = rstd;
      if (running_mean) {
        float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
    }
  }
  __syncthreads();
  if (c >= C) return;
  const float sc = s_scale[tx], sh = s_shift[tx];
  const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, M);
  for (int r = r0 + ty; r < r1; r += BN_RL) {
    float y = fmaf(X[(size_t)r * C + c], sc, sh);
    if (relu) y = fmaxf(y, 0.f);
    Y[(size_t)r * C + c] = y;
  }
}

// partial column sums of dz and dz*xhat (dz = dY gated by the fused ReLU)
__global__ void __launch_bounds__(256)
bn_bwd_partial_kernel(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ mean,
                      const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                      int relu, int M, int C, int rows_per_split, float* __restrict__ ws) {
  __shared__ float s_a[BN_RL][BN_COLS], s_b[BN_RL][BN_COLS];
  const int tx = threadIdx.x & (BN_COLS - 1), ty = threadIdx.x / BN_COLS;
  const int c = blockIdx.x * BN_COLS + tx;
  const int r0 = blockIdx.y * rows_per_split, r1 = min(r0 + rows_per_split, M);
  float sa = 0.f, sb = 0.f;
  if (c < C) {
    float mu = mean[c], rs = rstd[c], g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    for (int r = r0 + ty; r < r1; r += BN_RL) {
      float xh = (X[(size_t)r * C + c] - mu) * rs;
      float dz = dY[(size_t)r * C + c];
      if (relu && !(fmaf(xh, g, b) > 0.f)) dz = 0.f;
      sa += dz;
      sb = fmaf(dz, xh, sb);
    }
  }
  s_a[ty][tx] = sa; s_b[ty][tx] = sb;
  __syncthreads();
  if (ty == 0 && c < C) {
#pragma unroll
    for (int l = 1; l < BN_RL; ++l) { sa += s_a[l][tx]; sb += s_b[l][tx]; }
    float* o = ws + ((size_t)blockIdx.y * C + c) * 2;
    o[0] = sa; o[1] = sb;
  }
}

__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ mean,
                    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                    int relu, const float* __restrict__ ws, int M, int C, int splits, int rows_per_block,
                    float* __restrict__ dX, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float s_db[BN_COLS], s_dg[BN_COLS];
  __shared__ float s_pa[BN_RL][BN_COLS], s_pb[BN_RL][BN_COLS];
  const int tx = threadIdx.x & (BN_COLS - 1), ty = threadIdx.x / BN_COLS;
  const int c = blockIdx.x * BN_COLS + tx;
  {
    float pa = 0.f, pb = 0.f;
    if (c < C)
      for (int sidx = ty; sidx < splits; sidx += BN_RL) {
        const float* p = ws + ((size_t)sidx * C + c) * 2;
        pa += p[0]; pb += p[1];
      }
    s_pa[ty][tx] = pa; s_pb[ty][tx] = pb;
  }
  __syncthreads();
  if (ty == 0 && c < C) {
    float sa = ((s_pa[0][tx] + s_pa[1][tx]) + s_pa[2][tx]) + s_pa[3][tx];
    float sb = ((s_pb[0][tx] + s_pb[1][tx]) + s_pb[2][tx]) + s_pb[3][tx];
    s_db[tx] = sa; s_dg[tx] = sb;
    if (blockIdx.y == 0) {
      if (dbeta) dbeta[c] = sa;
      if (dgamma) dgamma[c] = sb;
    }
  }
  __syncthreads();
  if (c >= C) return;
  const float mu = mean[c], rs = rstd[c], g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float invM = 1.f / (float)M;
  const float kb = s_db[tx] * invM, kg = s_dg[tx] * invM;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, M);
  for (int r = r0 + ty; r < r1; r += BN_RL) {
    float xh = (X[(size_t)r * C + c] - mu) * rs;
    float dz = dY[(size_t)r * C + c];
    if (relu && !(fmaf(xh, g, b) > 0.f)) dz = 0.f;
    dX[(size_t)r * C + c] = g * rs * (dz - kb - xh * kg);
  }
}

static inline void bn_geometry(int M, int* splits, int* rows) {
  int s = (M + 127) / 128;          // >= 128 rows per split: (C/64) x splits workgroups fill the chip
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  *rows = (M + s - 1) / s;
  *splits = (M + *rows - 1) / *rows;
}

// ---- column sums (bias gradients of the library-GEMM weight-gradient path): two launches, fixed order
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float* __restrict__ X, int M, int C, int rows_per_split, float* __restrict__ ws) {
  __shared__ float s_a[BN_RL][BN_COLS];
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

extern "C" int msde_colsum(const float* X, int M, int C, float* out, float* workspace, void* stream) {
  if (M < 0 || C <= 0 || !X || !out || !workspace) return MSDE_EINVAL;
  hipStream_t st = as_stream(stream);
  if (M == 0) return (int)hipMemsetAsync(out, 0, (size_t)C * sizeof(float), st);
  int splits, rows;
  bn_geometry(M, &splits, &rows);
  MSDE_LAUNCH(colsum_partial_kernel, dim3((C + BN_COLS - 1) / BN_COLS, splits), dim3(256), 0, st, X, M, C, rows, workspace);
  MSDE_CHECK_LAUNCH();
  MSDE_LAUNCH(colsum_final_kernel, dim3((C + BN_COLS - 1) / BN_COLS), dim3(256), 0, st, (const float*)workspace, splits, C,
              out);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_bn_workspace_floats(int M, int C) {
  int splits, rows;
  bn_geometry(M, &splits, &rows);
  return splits * C * 3;
}

extern "C" int msde_bn_fwd(const float* X, int M, int C, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, int relu, float* Y,
                           float* save_mean, float* save_rstd, float* workspace, void* stream) {
  if (M <= 0 || C <= 0 || !X || !Y || !save_mean || !save_rstd || !workspace) return MSDE_EINVAL;
  int splits, rows;
  bn_geometry(M, &splits, &rows);
  dim3 grid((C + BN_COLS - 1) / BN_COLS, splits);
  MSDE_LAUNCH(bn_stats_partial_kernel, grid, dim3(256), 0, as_stream(stream), X, M, C, rows, workspace);
  MSDE_CHECK_LAUNCH();
  MSDE_LAUNCH(bn_fwd_apply_kernel, grid, dim3(256), 0, as_stream(stream), X, (const float*)workspace, M, C, splits, rows,
              gamma, beta, eps, momentum, running_mean, running_var, relu, Y, save_mean, save_rstd);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_bn_bwd(const float* dY, const float* X, const float* save_mean, const float* save_rstd,
                           const float* gamma, const float* beta, int relu, int M, int C, float* dX, float* dgamma,
                           float* dbeta, float* workspace, void* stream) {
  if (M <= 0 || C <= 0 || !dY || !X || !save_mean || !save_rstd || !dX || !workspace) return MSDE_EINVAL;
  int splits, rows;
  bn_geometry(M, &splits, &rows);
  dim3 grid((C + BN_COLS - 1) / BN_COLS, splits);
  MSDE_LAUNCH(bn_bwd_partial_kernel, grid, dim3(256), 0, as_stream(stream), dY, X, save_mean, save_rstd, gamma, beta, relu,
              M, C, rows, workspace);
  MSDE_CHECK_LAUNCH();
  MSDE_LAUNCH(bn_bwd_apply_kernel, grid, dim3(256), 0, as_stream(stream), dY, X, save_mean, save_rstd, gamma, beta, relu,
              (const float*)workspace, M, C, splits, rows, dX, dgamma, dbeta);
  MSDE_CHECK_LAUNCH();
  return 0;
}
